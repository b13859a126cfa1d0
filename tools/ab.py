"""Diagnostic: run bench.py against another build of the library (MZD_AB_SO=libmzd_x.so).  Not a benchmark of record."""
import os, sys, runpy
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), os.environ["MZD_AB_SO"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
