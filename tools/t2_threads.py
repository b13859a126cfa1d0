"""Diagnostic: mzd_decode_batch from several threads at once (each its own pinned output buffer): per-call times and the
aggregate rate -- what a multi-threaded caller (the daemon's batcher) sustains over PCIe.
  python tools/t2_threads.py [workload] [threads] [rounds]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, corpus, fuse_zstd_amd as mzd
from fuse_zstd_amd import api
mzd.init()
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kind, cfg, km, _ = bench.WORKLOADS[wl]
n = bench.DEFAULT_FILES[wl]
cp = corpus.build_corpus(kind, cfg, bench.file_sizes(wl, n, 0, 1), kind_mod=km)
end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
hin = mzd.HostBuffer(len(cp.comp)); hin.a[:] = cp.comp
outs = [mzd.HostBuffer(end + 64) for _ in range(nth)]
L = api.lib()
if os.environ.get("MZD_T2_MODE"): L.mzd_debug_host_path(0, 13, int(os.environ["MZD_T2_MODE"]))
jobs = [api.make_jobs([hin.a.ctypes.data + int(o) for o in cp.comp_offs], cp.comp_sizes, [o.a.ctypes.data + int(x) for x in cp.raw_offs], cp.raw_sizes) for o in outs]
L.mzd_decode_batch(jobs[0], n)
times = [[] for _ in range(nth)]
def worker(k):
    for _ in range(rounds):
        t0 = time.perf_counter()
        rc = L.mzd_decode_batch(jobs[k], n)
        times[k].append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
ths = [threading.Thread(target=worker, args=(k,)) for k in range(nth)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
for k in range(nth):
    assert bool((outs[k].a[:end] == cp.raw[:end]).all())
    print("thread %d: calls (ms) %s" % (k, " ".join("%.2f" % x for x in times[k])))
print("%d threads x %d calls: %.2f ms per batch, %.2f GiB/s decompressed in aggregate" % (nth, rounds, dt / (nth * rounds) * 1e3, nth * rounds * cp.raw_sizes.sum() / dt / 2**30))
