"""Stress (not a benchmark): the mutated multi-block batch of tests/test_gpu_parity.py::test_corrupted_multi_block_files_report_the_oracles_error
decoded again and again under each driver -- a status that differs from the oracle's, or from run to run (MZD_E_DEVICE = a bounded
wait ran out), would be a race in the error paths of the block tasks.   python tools/stress_corrupt.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuse_zstd_amd.api as api
if os.environ.get("MZD_SO"): api._SO = os.path.join(os.path.dirname(api._SO), os.environ["MZD_SO"])
import corpus, oracle, fuse_zstd_amd as mzd
mzd.init()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(77)
cases = []
for kind, size in (("json", 600000), ("text", 400000), ("xray", 300000), ("repeats", 500000)):
    cp = corpus.build_corpus(kind, 21, [size])
    good = cp.comp_file(0).tobytes()
    for _ in range(120):
        b = bytearray(good); pos = int(rng.randint(0, len(b))); b[pos] ^= int(rng.randint(1, 256)); cases.append((bytes(b), size))
    for cut in (len(good) - 1, len(good) - 5, len(good) // 2, 40): cases.append((good[:cut], size))
    for cap in (size - 1, size // 2, 150000, 10): cases.append((good, cap))
want = [oracle.decode(c, cap=cap)[0] for c, cap in cases]
bad_total = 0
for drv in [int(x) for x in os.environ.get("DRIVERS", "0,3,2,4,5,1").split(",")]:
    mzd.set_driver(drv)
    bad = 0
    for rep in range(reps):
        t0 = time.perf_counter()
        res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
        wall = time.perf_counter() - t0
        if os.environ.get("MZD_SITES"):
            first, top, cnt = mzd.debug_counters(0)[5:8]
            if cnt: print("driver", drv, "rep", rep, "wall %.2f s" % wall, "first: site %d job %d task %d; max: site %d job %d task %d; %d sites" % (first >> 16, (first >> 8) & 255, first & 255, top >> 16, (top >> 8) & 255, top & 255, cnt), flush=True)
        diff = [(i, st, want[i]) for i, (st, _) in enumerate(res) if (st == -6 if os.environ.get("MZD_SO") else st != want[i])]  # (an older build: only waits that ran out count -- its classes are another oracle's)
        if diff:
            bad += 1
            if bad <= 3: print("driver", drv, "rep", rep, diff[:6], "counter words of job 0's launch:", mzd.debug_counters(0), flush=True)
    print("driver %d: %d bad runs of %d" % (drv, bad, reps), flush=True)
    bad_total += bad
mzd.set_driver(0)
sys.exit(1 if bad_total else 0)
