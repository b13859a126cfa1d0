"""Diagnostic: per-workgroup role finish times of ONE file decoded by the block-task driver (libmzd_tfin.so), a row per task.
  python tools/tfin_rows.py [kind] [size] [driver]     -- cycles / 1000 since the task's block start"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), os.environ.get("MZD_DIAG_SO", "libmzd_tfin.so"))
import fuse_zstd_amd as mzd, corpus
import numpy as np
mzd.init()
kind = sys.argv[1] if len(sys.argv) > 1 else "json"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
drv = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cp = corpus.build_corpus(kind, 5, [size])
mzd.set_driver(drv)
for rep in range(3):
    res = mzd.decode_batch([cp.comp_file(0).tobytes()], [size])
assert res[0][0] == 0 and res[0][1] == cp.raw_file(0).tobytes()
buf = (C.c_uint64 * (12 * 2048))()
ns = api.lib().mzd_debug_tfin_all(0, buf, 2048)
arr = np.frombuffer(buf, dtype=np.uint64)[: ns * 12].reshape(ns, 12).astype(np.float64)
print("kernel ms", mzd.last_kernel_ms(0))
names = ["walk", "gather/copy", "hash", "plan", "jump(4)", "tables", "headers", "roles(7)", "build(8)", "pred(9)", "task->block", "task total"]
print(" ".join("%12s" % n for n in names))
for r in arr:
    if r[0] > 0: print(" ".join("%11.0fK" % (v / 1e3) for v in r))
