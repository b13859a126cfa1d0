"""Diagnostic: a few very big files through the host path under each driver (1 a workgroup per file, 5 / 4 block tasks without /
with blocks resolved ahead, 0 the library's own choice).  Kernel time only (PCIe excluded)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd as mzd, corpus
from fuse_zstd_amd import api as _api
if os.environ.get('MZD_AB_SO'): _api._SO = os.path.join(os.path.dirname(_api._SO), os.environ['MZD_AB_SO'])  # (another build of the library)
mzd.init()
for kind, size, n in (("json", 64 << 20, 1), ("text", 16 << 20, 4), ("xray", 32 << 20, 2)):
    cp = corpus.build_corpus(kind, 31, [size] * n)
    srcs = [cp.comp_file(i).tobytes() for i in range(n)]
    for drv in ("1", "5", "4", "0"):
        mzd.set_driver(int(drv))
        res = mzd.decode_batch(srcs, [size] * n)
        ok = all(st == 0 and out == cp.raw_file(i).tobytes() for i, (st, out) in enumerate(res))
        print(kind, n, "x", size >> 20, "MiB driver", drv, "kernel %.1f ms" % mzd.last_kernel_ms(0), "ok" if ok else "BAD %r" % [st for st, _ in res], flush=True)
