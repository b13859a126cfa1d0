"""Diagnostic: tools/t2_threads.py (cfg4, two threads, 60 calls each) with the host path's timeline on stderr (mzd_debug_host_path 7): a line
per phase of every mzd_decode_batch call -- where a call's wall time goes when two calls are in flight (profiles/r06_t2_pairs.txt)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.argv = ["t2_threads.py"] + (sys.argv[1:] or ["cfg4", "2", "60"])
import fuse_zstd_amd as mzd
mzd.init()
mzd.lib().mzd_debug_host_path(0, 7, 1)
exec(open(os.path.join(root, "tools", "t2_threads.py")).read().replace("mzd.init()", ""))
