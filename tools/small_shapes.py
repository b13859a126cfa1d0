"""Diagnostic: kernel time of N small JSON files under the general driver and every shape of the small-file kernel (files per wavefront /
files executed at a time) -- the data behind make_plan's choice of shape.   python tools/small_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import corpus, fuse_zstd_amd as mzd
import fuse_zstd_amd.api as api
mzd.init()
dev = torch.device("cuda:0")
L = api.lib()
SHAPES = ((4, 4), (8, 4), (4, 2), (8, 8), (16, 16))
def run(jobs, reps=3):
    best = 1e9
    for _ in range(reps):
        res = mzd.decode_batch_device(0, jobs)
        assert all(st == 0 for st, _ in res)
        best = min(best, mzd.last_kernel_ms(0))
    return best
print("%-6s %-6s %8s %-28s %8s " % ("size", "files", "auto", "(kernel)", "general") + " ".join("%7s" % ("%d/%d" % s) for s in SHAPES))
for size in (512, 1024, 2048, 3072, 4096, 6144, 8192):
    for n in (2048, 10000, 40000):
        if size * n > 170 << 20: continue
        cp = corpus.build_corpus("json", 4, [size] * n)
        comp = torch.from_numpy(cp.comp).to(dev)
        end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
        out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
        jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
        mzd.set_driver(0); auto = run(jobs); name = mzd.last_kernel_name(0)
        mzd.set_driver(1); gen = run(jobs); mzd.set_driver(3)
        ts = []
        for g, xg in SHAPES:
            L.mzd_debug_host_path(0, 4, g); L.mzd_debug_host_path(0, 5, xg)
            ts.append(run(jobs))
        L.mzd_debug_host_path(0, 4, 0); L.mzd_debug_host_path(0, 5, 0); mzd.set_driver(0)
        print("%-6d %-6d %8.3f %-28s %8.3f " % (size, n, auto, name, gen) + " ".join("%7.3f" % t for t in ts), flush=True)
