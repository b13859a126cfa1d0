"""Diagnostic: what dictionary content resident in LDS could save, and what it would cost, measured with the shipped
kernel: (a) a build that skips the trips to the content in L2 (-DMZD_EXP_NODICTFETCH: bytes wrong, time an upper bound of the saving);
(b) the shipped build with only as many wavefronts resident as would fit beside 112 KB of content (GRID).  python tools/dict_lds_bound.py"""
import os, sys, subprocess
CH = r'''
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), sys.argv[1])
import corpus, fuse_zstd_amd as mzd
mzd.init()
if os.environ.get('G'): api.lib().mzd_debug_host_path(0, 4, int(os.environ['G']))
if os.environ.get('GRID'): api.lib().mzd_debug_host_path(0, 6, int(os.environ['GRID']))
dev = torch.device("cuda:0")
rs = np.random.RandomState(55).randint(300, 3001, size=50000)
d = corpus.train_dict("json", 5, [int(x) for x in rs[:4000]], cap=112640)
h = mzd.load_dict(d)
cp = corpus.build_corpus("json", 5, [int(x) for x in rs], dictionary=d)
comp = torch.from_numpy(cp.comp).to(dev)
end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
out = torch.zeros(end + 64, dtype=torch.uint8, device=dev)
jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes, [h] * cp.nfiles)
ts = []
for _ in range(7):
    res = mzd.decode_batch_device(0, jobs); ts.append(mzd.last_kernel_ms(0))
print("%-16s G=%s grid=%s: cfg5 kernel best %.3f ms, median %.3f ms" % (sys.argv[1], os.environ.get("G", "auto(8)"), os.environ.get("GRID", "all"), min(ts), sorted(ts)[3]), flush=True)
'''
for so, env in [("libmzd.so", {}), ("libmzd_exp.so", {}), ("libmzd.so", {"GRID": "256"}), ("libmzd.so", {"GRID": "256", "G": "4"}), ("libmzd.so", {"GRID": "512", "G": "4"}), ("libmzd_exp.so", {"GRID": "256"}), ("libmzd.so", {})]:
    subprocess.run([sys.executable, "-c", CH, so], env=dict(os.environ, **env))
