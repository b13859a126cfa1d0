"""Diagnostic: kernel time of N small JSON files under each kernel choice -- the general driver alone (mode 1), the LDS kernel
(mode 3), and -- while it existed: mode 6, removed after this measurement -- the round-2 lane-per-file kernel -- per file size and file count: the data behind the launch policy
(mzd_host.cpp: make_plan) and the record asked for by "measure LDS-window vs before per size class".
  python tools/small_policy.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import corpus, fuse_zstd_amd as mzd
import fuse_zstd_amd.api as api
mzd.init()
dev = torch.device("cuda:0")

def t(cp, mode, reps=4):
    comp = torch.from_numpy(cp.comp).to(dev)
    end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
    out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
    jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
    mzd.set_driver(mode)
    best = 1e9
    for _ in range(reps):
        res = mzd.decode_batch_device(0, jobs)
        assert all(st == 0 for st, _ in res)
        best = min(best, mzd.last_kernel_ms(0))
    mzd.set_driver(0)
    return best

print("%-8s %-7s %12s %12s %12s   (kernel ms; GiB/s of the best)" % ("size", "files", "general", "LDS", "lane/file"))
for size in (512, 1024, 2048, 4096, 8192):
    for n in (64, 256, 1024, 2048, 4096, 10000, 40000):
        if size * n > 200 << 20: continue
        cp = corpus.build_corpus("json", 4, [size] * n)
        a, b = t(cp, 1), t(cp, 3); c = float("nan")  # (mode 6 is gone: see profiles/r03_small_policy.txt for its last measurement)
        print("%-8d %-7d %12.3f %12.3f %12.3f   %.1f" % (size, n, a, b, c, size * n / min(a, b, c) / 1e-3 / 2**30), flush=True)
