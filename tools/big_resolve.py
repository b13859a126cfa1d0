"""Diagnostic: n multi-block JSON files of one size, device-resident, under the block-task driver with its three ways of executing a block
(mzd_debug_host_path 10: 1 in order, 2 every task resolved ahead, 3 only tasks whose predecessor is still running, 4 every other task) -- kernel ms.
  python tools/big_resolve.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import corpus, fuse_zstd_amd as mzd
import fuse_zstd_amd.api as api
mzd.init()
dev = torch.device("cuda:0")
L = api.lib()
print("%-8s %-6s %10s %10s %10s %10s %10s" % ("size", "files", "auto", "in order", "all ahead", "behind", "odd ahead"))
SWEEP = [(1 << 20, int(x)) for x in os.environ["MZD_SWEEP"].split(",")] if os.environ.get("MZD_SWEEP") else [(1 << 20, 1), (1 << 20, 16), (1 << 20, 100), (1 << 20, 200), (1 << 20, 400), (1 << 20, 800), (1 << 18, 1600), (1 << 22, 100)]
for size, n in SWEEP:
    cp = corpus.build_corpus("json", 1, [size] * n)
    comp = torch.from_numpy(cp.comp).to(dev)
    end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
    out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
    jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
    ts = []
    for mode in (0, 1, 2, 3, 4):
        L.mzd_debug_host_path(0, 10, mode)
        best = 1e9
        for _ in range(3):
            res = mzd.decode_batch_device(0, jobs)
            assert all(st == 0 for st, _ in res)
            best = min(best, mzd.last_kernel_ms(0))
        ts.append(best)
    L.mzd_debug_host_path(0, 10, 0)
    print("%-8d %-6d %10.3f %10.3f %10.3f %10.3f %10.3f   %s" % (size, n, ts[0], ts[1], ts[2], ts[3], ts[4], mzd.last_kernel_name(0)), flush=True)
