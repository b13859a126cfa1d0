"""Diagnostic: driver 1 with one file a workgroup and with two (mzd_debug_host_path 11), n files of 128 KiB, kernel ms side by side.
python tools/pairs_probe.py [kind] [n ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as _api
if os.environ.get('MZD_AB_SO'): _api._SO = os.path.join(os.path.dirname(_api._SO), os.environ['MZD_AB_SO'])
import fuse_zstd_amd as mzd, corpus
import torch
mzd.init()
kind = sys.argv[1] if len(sys.argv) > 1 else "json"
for n in [int(x) for x in sys.argv[2:]] or [256, 512, 1000, 2000, 4000, 8000]:
    cp = corpus.build_corpus(kind, 2, [131072] * n)
    dcomp = torch.from_numpy(cp.comp).cuda()
    end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
    dout = torch.zeros(end + 64, dtype=torch.uint8, device="cuda")
    jobs = mzd.api.make_jobs([dcomp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [dout.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
    row = []
    for way in [int(x) for x in os.environ.get("MZD_WAYS", "1,2,1,2").split(",")]:
        if os.environ.get("MZD_DIAG_EACH") and row: mzd.debug_counters(0)
        mzd.lib().mzd_debug_host_path(0, 11, way)
        best = 1e9
        for rep in range(3):
            dout.zero_(); torch.cuda.synchronize()
            res = mzd.api.decode_batch_device(0, jobs)
            best = min(best, mzd.last_kernel_ms(0))
        ok = all(st == 0 for st, _ in res) and bytes(dout.cpu().numpy()[:end]) == cp.raw[:end].tobytes()
        row.append("%s %.3f%s" % ({1: "four-wave", 2: "pairs", 3: "three-wave"}[way], best, "" if ok else " WRONG"))
    mzd.lib().mzd_debug_host_path(0, 11, 0)
    print(kind, n, "files:", "  ".join(row), flush=True)
    mzd.debug_counters(0)  # (experiment builds print what they noted)
