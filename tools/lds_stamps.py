"""Diagnostic: phase times of the LDS small-file kernel (mzd_lds.hip), from a library built with `make sstamps`:
cycles between the phase boundaries of workgroup 0's first group, and the kernel time of the launch.
  python tools/lds_stamps.py [cfg4|cfg5] [files]      (G=4|8|16 [XG=4] in the environment of THIS tool pick the files per wavefront / executed at a time: mzd_debug_host_path 4 / 5)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), os.environ.get("MZD_SO", "libmzd_sstamps.so"))
import bench, corpus, fuse_zstd_amd as mzd
mzd.init()
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_FILES[wl]
kind, cfg, km, _ = bench.WORKLOADS[wl]
sizes = bench.file_sizes(wl, n, 0, 1)
d, did = None, 0
if wl == "cfg5":
    tr = np.random.RandomState(55).randint(300, 3001, size=4000)
    d = corpus.train_dict(kind, cfg, [int(x) for x in tr], cap=112640)
    did = mzd.load_dict(d)
cp = corpus.build_corpus(kind, cfg, sizes, kind_mod=km, dictionary=d)
import torch
dev = torch.device("cuda:0")
comp = torch.from_numpy(cp.comp).to(dev)
end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes,
                     [did] * n if did else None)
torch.cuda.synchronize()
names = ["take group, job entries, input -> LDS", "headers", "Huffman weights + table", "Huffman streams", "sequence header, counts",
         "FSE tables", "walk + extract -> scratch", "execute (window)", "XXH64", "flush + results"]
mzd.set_driver(3)
if os.environ.get('G'): api.lib().mzd_debug_host_path(0, 4, int(os.environ['G']))
if os.environ.get('XG'): api.lib().mzd_debug_host_path(0, 5, int(os.environ['XG']))
for rep in range(3):
    res = mzd.decode_batch_device(0, jobs)
    assert all(st == 0 for st, _ in res)
    st = (C.c_uint64 * 1032)()
    api.lib().mzd_debug_small_stamps.argtypes = [C.c_int, C.c_void_p]
    api.lib().mzd_debug_small_stamps(0, st)
    t = list(st)
    if t[64] and rep == 2:
        import json
        json.dump([(x >> 32, x & 0xFFFFFFFF) for x in t[65:65 + min(t[64], 960)]], open("gpurun_out/lds_failed.json", "w"))
    if any(t[32:48]): print("    left the fast path, by reason 0..15:", t[32:48], "last (file, group<<8|lane):", [(x >> 32, hex(x & 0xFFFFFFFF)) for x in t[48:64] if x])
    if rep == 2:
        print("%s G=%s: kernel %.3f ms; workgroup 0, first group: total %d cycles" % (wl, os.environ.get("G", "auto") + "/" + os.environ.get("XG", "-"), mzd.last_kernel_ms(0), t[9] - t[0]))
        print("    inside: walk %d, extract %d cycles" % (t[10], t[11]))
        if t[13] and t[14] and t[15]: print("    Huffman weights: counts %d, their FSE table %d, the weights %d, validation + decode table %d cycles" % (t[13] - t[2], t[14] - t[13], t[15] - t[14], t[3] - t[15]))
        print("    execute: stage A %d, repeat offsets %d, literals %d, match rounds %d cycles (%d rounds)" % (t[18], t[19], t[20], t[21], t[22]))
        print("    inside the rounds: first waiting sequence %d, copies %d, rare matches + loop %d cycles (%d rare calls)" % (t[23], t[24], t[25], t[26]))
        t = t[:8] + [t[17]] + t[8:10] + t[10:]  # (the execute stamp was added later: index 17)
        for k in range(10):
            print("    %-46s %8d cycles" % (names[k], t[k + 1] - t[k]))
