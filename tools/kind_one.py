"""Diagnostic: kernel time of N files of ONE corpus kind (device-resident, one launch); for rocprofv3 runs.   python tools/kind_one.py xray 1000 [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as _api
if os.environ.get('MZD_AB_SO'): _api._SO = os.path.join(os.path.dirname(_api._SO), os.environ['MZD_AB_SO'])
import fuse_zstd_amd as mzd, corpus
import torch
mzd.init()
if os.environ.get("MZD_DRIVER"): mzd.set_driver(int(os.environ["MZD_DRIVER"]))  # (mzd_debug_set_driver: 1 a workgroup per file, 2 block tasks, 4 ... all resolved ahead, 5 ... none)
kind = sys.argv[1]; n = int(sys.argv[2]); size = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
cp = corpus.build_corpus(kind, 3, [size] * n)
dcomp = torch.from_numpy(cp.comp).cuda()
dout = torch.zeros(int(cp.raw_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=torch.uint8, device="cuda")
jobs = mzd.api.make_jobs([dcomp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [dout.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
ms = []
for rep in range(6):
    mzd.api.decode_batch_device(0, jobs); torch.cuda.synchronize(); ms.append(mzd.last_kernel_ms(0))
got = dout.cpu().numpy()
ok = all(j.status == 0 for j in jobs) and all(bytes(got[int(cp.raw_offs[i]):int(cp.raw_offs[i]) + size]) == cp.raw_file(i).tobytes() for i in range(0, n, 17))
print("%s x %d: ok=%s kernel ms %s (%s)" % (kind, n, ok, " ".join("%.3f" % m for m in ms), mzd.last_kernel_name(0)), flush=True)
print("workgroups of the last launch (counter word 6):", mzd.debug_counters(0)[6], flush=True)
