"""Diagnostic: phase times of the small-file kernel (mzd_small.hip), from a library built with `make sstamps`:
cycles between the phase boundaries of workgroup 0's first group, and the kernel time of the launch.
  python tools/small_stamps.py [cfg4|cfg5] [files]"""
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
names = ["take group + headers (A)", "Huffman tree staging, weights, table (B, C)", "Huffman streams (D)", "sequence header staging + parse (E)",
         "FSE tables (F)", "bitstream staging, walk + execute (G)", "results to LDS", "XXH64 (H)"]
for rep in range(3):
    res = mzd.decode_batch_device(0, jobs)
    if not os.environ.get("MZD_SO"): assert all(st == 0 for st, _ in res)
    st = (C.c_uint64 * 12)()
    api.lib().mzd_debug_small_stamps.argtypes = [C.c_int, C.c_void_p]
    api.lib().mzd_debug_small_stamps(0, st)
    t = list(st)
    print("pass %d: kernel %.3f ms; workgroup 0, first group: total %d cycles" % (rep, mzd.last_kernel_ms(0), t[8] - t[0]))
    print("    inside G: requests %d, walk of the next step %d, stores %d cycles" % (t[9], t[10], t[11]))
    for k in range(8):
        print("    %-46s %8d cycles" % (names[k], t[k + 1] - t[k]))
