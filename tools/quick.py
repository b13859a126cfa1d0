"""Diagnostic: decode named golden vectors one by one (prints as it goes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd as mzd
from tests import golden_util
mzd.init()
vs = golden_util.load_manifest()
names = sys.argv[1:] or [v.name for v in vs if v.dict is None]
for nm in names:
    v = next(x for x in vs if x.name == nm)
    print(nm, end=" ", flush=True)
    st, out = mzd.decode(v.comp, v.out_len if v.ok else 1 << 22)
    print("status", st, "ok", (out == v.expected()) if v.ok else ("expect", v.oracle_class), flush=True)
