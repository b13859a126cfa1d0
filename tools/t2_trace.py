import os, sys
sys.path.insert(0, "/root/repo")
sys.argv = ["t2_threads.py", "cfg4", "2", "60"]
import fuse_zstd_amd as mzd
mzd.init()
mzd.lib().mzd_debug_host_path(0, 7, 1)
exec(open("/root/repo/tools/t2_threads.py").read().replace("mzd.init()", ""))
