"""Diagnostic: mutated frames WITHOUT a checksum (seed = argv[1]) -- what the reference's libzstd 1.5 accepts although a literal stream is not consumed
exactly must come back with the oracle's bytes (oracle/zstd_oracle.c: g_huf_rule) -- eight data classes x three levels, one to seven mutated bytes
near each other, decoded on the GPU under the library's choice and under each general driver; exits non-zero on any mismatch.
tests/test_gpu_parity.py::test_literal_streams_not_consumed_exactly_follow_the_pinned_libzstd is the committed, smaller form."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import corpus, oracle, fuse_zstd_amd as mzd
mzd.init()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
rng = np.random.RandomState(seed)
Z = oracle.LibZstd
cases = []
for kind, size in (("json", 131072), ("text", 100000), ("markup", 60000), ("xray", 131072), ("json", 20000), ("dna", 50000), ("int32", 131072), ("json", 4096), ("text", 300000), ("xray", 400000)):
    for level in (1, 3, 19):
        if level == 19 and size > 140000: continue
        good = Z.compress(corpus.gen(kind, seed, 1, size), level, False)
        for _ in range(100):
            b = bytearray(good)
            pos = int(rng.randint(6, len(b)))
            for _ in range(int(rng.randint(1, 8))):
                b[min(len(b) - 1, pos + int(rng.randint(0, 1500)))] ^= int(rng.randint(1, 256))
            cases.append((bytes(b), size))
want = []
lenient = through = 0
for comp, cap in cases:
    rc, out = oracle.decode(comp, cap=cap)
    if rc == 0: lenient += oracle.last_verdict_lit_lenient(); through += oracle.last_verdict_lit_through()
    want.append((rc, out))
bad = 0
for drv in (0, 1, 4, 5):
    mzd.set_driver(drv)
    res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
    nb = sum(1 for (st, out), (rc, w) in zip(res, want) if st != rc or (st == 0 and out != w))
    print("driver", drv, "cases", len(cases), "bad", nb, "accepted", sum(1 for st, _ in res if st == 0), "(leftover bits ignored %d, read through %d)" % (lenient, through), flush=True)
    bad += nb
mzd.set_driver(0)
sys.exit(1 if bad else 0)
