"""Diagnostic: the small-file workloads' kernel time under several builds of the library in ONE GPU call (boxes differ by up to 10 %,
so builds are only comparable side by side).   python tools/lds_ab.py libmzd.so libmzd_prev.so [...]      (each build runs in its own process, twice, alternating)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), sys.argv[1])
import corpus, fuse_zstd_amd as mzd
mzd.init()
dev = torch.device("cuda:0")
def t(cp, did=0, reps=7):
    comp = torch.from_numpy(cp.comp).to(dev)
    end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
    out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
    jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes, [did] * cp.nfiles if did else None)
    ts = []
    for _ in range(reps):
        res = mzd.decode_batch_device(0, jobs)
        assert all(st == 0 for st, _ in res)
        ts.append(mzd.last_kernel_ms(0))
    return min(ts), sorted(ts)[len(ts) // 2]
a = t(corpus.build_corpus("json", 4, [4096] * 10000))
b = t(corpus.build_corpus("json", 4, [4096] * 40000))
rs = np.random.RandomState(55).randint(300, 3001, size=50000)
d = corpus.train_dict("json", 5, [int(x) for x in rs[:4000]], cap=112640)
h = mzd.load_dict(d)
c = t(corpus.build_corpus("json", 5, [int(x) for x in rs], dictionary=d), h)
print("%%-22s cfg4 %%.3f (median %%.3f)  cfg4x4 %%.3f (%%.3f)  cfg5 %%.3f (%%.3f) ms" %% (sys.argv[1], a[0], a[1], b[0], b[1], c[0], c[1]), flush=True)
''' % ROOT
for rnd in range(2):
    for so in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, so], check=False)
