"""Stress of the block tasks' hand-over between workgroups (not a benchmark): every positive golden vector without a dictionary in ONE
batch, decoded again and again under the library's own choice of driver.  `window_log10` (586 blocks of 1 KiB, each resolved ahead of
its predecessor, the checksum chain's state travelling from task to task on its own flag) is what found a missing agent-scope release
in front of the hand-over flags: right bytes, wrong digest, about once in a hundred runs.  Exits non-zero on any mismatch.
  python tools/stress_handover.py [runs] [library]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as api
if len(sys.argv) > 2: api._SO = os.path.join(os.path.dirname(api._SO), sys.argv[2])
import fuse_zstd_amd as mzd
from tests import golden_util
mzd.init()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
vs = [v for v in golden_util.load_manifest() if v.ok and v.dict is None]
bad_runs = 0
for rep in range(runs):
    res = mzd.decode_batch([v.comp for v in vs], [v.out_len for v in vs])
    bad = [(v.name, st) for v, (st, out) in zip(vs, res) if st != 0 or out != v.expected()]
    if bad:
        bad_runs += 1
        if bad_runs <= 5: print("run", rep, bad, flush=True)
print("%s: %d bad runs of %d" % (os.path.basename(api._SO), bad_runs, runs), flush=True)
sys.exit(1 if bad_runs else 0)
