"""Diagnostic: random mutations (seed = argv[1]; 1..3 bytes each), truncations and too-small outputs of MULTI-BLOCK frames of four
data classes, decoded by the block-task driver with blocks resolved ahead (driver 4), without (5) and by the library's own choice
(0) and by driver 1 (a workgroup per file), and compared with the oracle (status, and bytes where both accept); exits non-zero on any mismatch.
tests/test_gpu_parity.py::test_corrupted_multi_block_files_report_the_oracles_error is the committed, smaller form."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import corpus, oracle, fuse_zstd_amd as mzd
mzd.init()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
rng = np.random.RandomState(seed)
cases = []
for kind, size in (("json", 600000), ("text", 400000), ("xray", 300000), ("repeats", 500000), ("int32", 350000), ("markup", 280000)):
    for level in (1, 3, 9):
        cp = corpus.build_corpus(kind, 21 + seed, [size], level=level)
        good = cp.comp_file(0).tobytes()
        for _ in range(60):
            b = bytearray(good)
            for _ in range(int(rng.randint(1, 4))):
                b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
            cases.append((bytes(b), size))
        for cut in (len(good) - 1, len(good) - 5, len(good) // 2, int(rng.randint(1, len(good)))):
            cases.append((good[:cut], size))
        for cap in (size - 1, size // 2, 150000, int(rng.randint(1, size))):
            cases.append((good, cap))
        cases.append((good, size))
want = [oracle.decode(c, cap=cap) for c, cap in cases]
bad = 0
for drv in (4, 5, 0, 1):  # (1: a workgroup per file -- what a launch that fills the machine gets for multi-block files too)
    mzd.set_driver(drv)
    res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
    for i, ((st, out), (rc, ref)) in enumerate(zip(res, want)):
        if st != rc or (st == 0 and out != ref):
            bad += 1
            print("MISMATCH driver", drv, "case", i, st, rc)
    print("driver", drv, "cases", len(cases), "accepted", sum(1 for st, _ in res if st == 0), flush=True)
sys.exit(1 if bad else 0)
