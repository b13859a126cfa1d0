"""Diagnostic (T2 of SURVEY.md 8d): wall time of mzd_decode_batch on HOST buffers (the chunked pipeline of mzd_host.cpp:
copy in | decode | copy out on separate streams), i.e. the PCIe-inclusive rate of a corpus.  Not the bench's headline value.
  python tools/t2_host_path.py [workload] [files] [pinned|pageable]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, corpus
from fuse_zstd_amd import api
if os.environ.get('MZD_AB_SO'): api._SO = os.path.join(os.path.dirname(api._SO), os.environ['MZD_AB_SO'])  # (another build of the library)
import fuse_zstd_amd as mzd
mzd.init()
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_FILES[wl]
pinned = (sys.argv[3] if len(sys.argv) > 3 else "pinned") == "pinned"
kind, cfg, km, _ = bench.WORKLOADS[wl]
cp = corpus.build_corpus(kind, cfg, bench.file_sizes(wl, n, 0, 1), kind_mod=km)
end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
if pinned:
    hin = mzd.HostBuffer(len(cp.comp)); hin.a[:] = cp.comp; src = hin.a
    hout = mzd.HostBuffer(end + 64); out = hout.a
else:
    src = cp.comp; out = np.zeros(end + 64, dtype=np.uint8)
jobs = api.make_jobs([src.ctypes.data + int(o) for o in cp.comp_offs], cp.comp_sizes,
                     [out.ctypes.data + int(o) for o in cp.raw_offs], cp.raw_sizes)
L = api.lib()
if os.environ.get('MZD_COPY_THREADS'): L.mzd_debug_host_path(0, 3, int(os.environ['MZD_COPY_THREADS']))
if os.environ.get('MZD_DIRECT_CHUNKS'): L.mzd_debug_host_path(0, 2, int(os.environ['MZD_DIRECT_CHUNKS']))
for rep in range(6):
    if not os.environ.get('MZD_NOZERO'): out[:end] = 0
    t0 = time.perf_counter()
    rc = L.mzd_decode_batch(jobs, n)
    dt = time.perf_counter() - t0
    bad = [(i, j.status) for i, j in enumerate(jobs) if j.status != 0]
    assert rc == 0 and not bad, (rc, len(bad), bad[:8], bad[-4:])
    ok = bool((out[:end] == cp.raw[:end]).all())
    print("pass %d: %.3f ms wall, %.2f GiB/s decompressed (host -> host, %s), kernels %.3f ms summed over chunks, bytes ok %s" % (
        rep, dt * 1e3, cp.raw_sizes.sum() / dt / 2**30, "pinned" if pinned else "pageable", mzd.last_kernel_ms(0), ok), flush=True)
