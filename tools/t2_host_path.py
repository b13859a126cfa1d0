"""Diagnostic (T2 of SURVEY.md 8d): wall time of mzd_decode_batch on HOST buffers (pinned staging -> H2D -> kernel -> D2H),
i.e. the PCIe-inclusive rate of the cfg2 corpus.  Not the bench's headline value."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corpus, fuse_zstd_amd as mzd
from fuse_zstd_amd import api
mzd.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cp = corpus.build_corpus("json", 2, [131072] * n)
out = np.zeros(int(cp.raw_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=np.uint8)
jobs = api.make_jobs([cp.comp.ctypes.data + int(o) for o in cp.comp_offs], cp.comp_sizes,
                     [out.ctypes.data + int(o) for o in cp.raw_offs], cp.raw_sizes)
L = api.lib()
for rep in range(6):
    t0 = time.perf_counter()
    rc = L.mzd_decode_batch(jobs, n)
    dt = time.perf_counter() - t0
    assert rc == 0 and all(j.status == 0 for j in jobs)
    end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
    ok = bool((out[:end] == cp.raw[:end]).all())
    print("pass %d: %.3f ms wall, %.2f GiB/s decompressed (host -> host), kernel %.3f ms, bytes ok %s" % (
        rep, dt * 1e3, cp.raw_sizes.sum() / dt / 2**30, mzd.last_kernel_ms(0), ok), flush=True)
