import sys, time
sys.path.insert(0,'.')
import fuse_zstd_amd as mzd, oracle
from tests import golden_util
mzd.init()
vs=golden_util.load_manifest()
names=sys.argv[1:] or ["ref_bulk_01","ref_writer_00","json_4k"]
for nm in names:
    v=next(x for x in vs if x.name==nm)
    t=time.time()
    st,out=mzd.decode(v.comp, v.out_len if v.ok else 1<<22)
    print(nm, "status",st, "len",len(out), "ok", (out==v.expected()) if v.ok else ("expect",v.oracle_class), "%.3fs"%(time.time()-t), flush=True)
