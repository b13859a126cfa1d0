"""Diagnostic: the LDS small-file kernel (mzd_lds.hip) on corpora of small files, against the generator's bytes.
Prints per corpus: wrong files, files handed on to the general driver (counter word 4), kernel time.
  python tools/lds_check.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import corpus, fuse_zstd_amd as mzd
import fuse_zstd_amd.api as api
if os.environ.get('MZD_SO'): api._SO = os.path.join(os.path.dirname(api._SO), os.environ['MZD_SO'])

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
mzd.init()
dev = torch.device("cuda:0")


def run(name, cp, did=0, mode=3, reps=3):
    comp = torch.from_numpy(cp.comp).to(dev)
    end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
    out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
    n = cp.nfiles
    jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes,
                         [did] * n if did else None)
    torch.cuda.synchronize()
    mzd.set_driver(mode)
    best = 1e9
    for _ in range(reps):
        out.zero_()
        torch.cuda.synchronize()
        res = mzd.decode_batch_device(0, jobs)
        best = min(best, mzd.last_kernel_ms(0))
    mzd.set_driver(0)
    c = mzd.debug_counters(0)
    got = out.cpu().numpy()
    wrong, badst = [], []
    for i, (st, ln) in enumerate(res):
        o, sz = int(cp.raw_offs[i]), int(cp.raw_sizes[i])
        if st != 0 or ln != sz: badst.append((i, st, ln, sz))
        elif not np.array_equal(got[o:o + sz], cp.raw[o:o + sz]): wrong.append(i)
    U = int(cp.raw_sizes.sum())
    print("%-34s files %6d  wrong %d  bad status %d  handed on %d  groups %d  kernel %.3f ms  %.1f GiB/s" % (
        name, n, len(wrong), len(badst), c[4], c[5], best, U / best / 1e-3 / 2**30), flush=True)
    if wrong[:5]:
        i = wrong[0]; o, sz = int(cp.raw_offs[i]), int(cp.raw_sizes[i])
        d = np.nonzero(got[o:o + sz] != cp.raw[o:o + sz])[0]
        print("   first wrong files", wrong[:8], "file", i, "size", sz, "first diffs at", d[:10], "ndiff", len(d))
        print("   got ", bytes(got[o + int(d[0]) - 8:o + int(d[0]) + 40]))
        print("   want", bytes(cp.raw[o + int(d[0]) - 8:o + int(d[0]) + 40]))
        print("   all diff positions of this file:", [int(x) for x in d[:64]])
    if badst[:5]: print("   bad status (file, status, len, want):", badst[:8])
    return len(wrong) + len(badst)


bad = 0
bad += run("json 4 KiB x 64", corpus.build_corpus("json", 4, [4096] * 64))
if not quick:
    sizes = [0, 1, 2, 7, 15, 16, 17, 31, 32, 33, 63, 64, 100, 255, 256, 300, 511, 700, 1000, 1023, 1024, 2000, 3000, 4095, 4096, 4097, 5000, 6000, 8191, 8192]
    for kind in ["json", "text", "markup", "int32", "dna", "xray", "random", "repeats"]:
        for level in (1, 3, 19):
            bad += run("%s level %d sizes 0..8192" % (kind, level), corpus.build_corpus(kind, 77, sizes * 3, level=level), reps=1)
bad += run("cfg4: json 4 KiB x 10000", corpus.build_corpus("json", 4, [4096] * 10000))
rs = np.random.RandomState(55).randint(300, 3001, size=50000)
d = corpus.train_dict("json", 5, [int(x) for x in rs[:4000]], cap=112640)
h = mzd.load_dict(d)
nf = 2000 if quick else 50000
bad += run("cfg5: dict records x %d" % nf, corpus.build_corpus("json", 5, [int(x) for x in rs[:nf]], dictionary=d), did=h)
print("TOTAL BAD", bad)
