"""Diagnostic: the log-uniform mix (cfg4lu) under the two general drivers, files in the caller's order and largest first.
  python tools/lpt_order.py        (kernel ms of one whole-device launch, best of 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, corpus, fuse_zstd_amd as mzd
import fuse_zstd_amd.api as api
mzd.init()
dev = torch.device("cuda:0")
kind, cfg, km, _ = bench.WORKLOADS["cfg4lu"]
sizes = bench.file_sizes("cfg4lu", bench.DEFAULT_FILES["cfg4lu"], 0, 1)
cp = corpus.build_corpus(kind, cfg, sizes, kind_mod=km)
comp = torch.from_numpy(cp.comp).to(dev)
end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
U = float(sum(sizes))
for order_name, order in (("caller's order", np.arange(len(sizes))), ("largest first", np.argsort(-np.array(sizes), kind="stable"))):
    jobs = api.make_jobs([comp.data_ptr() + int(cp.comp_offs[i]) for i in order], [cp.comp_sizes[i] for i in order],
                         [out.data_ptr() + int(cp.raw_offs[i]) for i in order], [cp.raw_sizes[i] for i in order])
    for drv in (0, 1):
        mzd.set_driver(drv)
        ts = []
        for _ in range(5):
            res = mzd.decode_batch_device(0, jobs)
            assert all(st == 0 for st, _ in res)
            ts.append(mzd.last_kernel_ms(0))
        print("%-16s driver %d (%s): %.3f ms = %.1f GiB/s" % (order_name, drv, "library's choice" if drv == 0 else "a workgroup per file", min(ts), U / (min(ts) * 1e-3) / 2**30), flush=True)
mzd.set_driver(0)
