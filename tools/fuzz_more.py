"""Diagnostic: random multi-byte mutations (seed = argv[1]) of frames of eight data classes x three levels, decoded on the
GPU in one launch and compared with the oracle (status, and bytes where both accept); exits non-zero on any mismatch.
tests/test_gpu_parity.py::test_multi_byte_mutations_match_oracle is the committed, smaller form."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import corpus, oracle, fuse_zstd_amd as mzd
mzd.init()
if os.environ.get('MZD_DRIVER'): mzd.set_driver(int(os.environ['MZD_DRIVER']))  # (3: the small-file kernel for the small frames)
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
cases = []
for kind, seed, size in (("json", 41, 131072), ("text", 42, 100000), ("markup", 43, 60000), ("xray", 44, 131072), ("json", 45, 20000), ("dna", 46, 50000), ("repeats", 47, 131072), ("json", 48, 4096)):
    for level in (1, 3, 19):
        cp = corpus.build_corpus(kind, seed, [size], level=level)
        good = cp.comp_file(0).tobytes()
        for _ in range(150):
            b = bytearray(good)
            for _ in range(int(rng.randint(1, 4))):
                b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
            cases.append((bytes(b), size))
res = mzd.decode_batch([c for c, _ in cases], [cap for _, cap in cases])
bad = 0
for i, ((comp, cap), (st, out)) in enumerate(zip(cases, res)):
    rc, want = oracle.decode(comp, cap=cap)
    if st != rc or (st == 0 and out != want):
        bad += 1
        print("MISMATCH", i, st, rc)
print("cases", len(cases), "bad", bad, "accepted", sum(1 for st, _ in res if st == 0))
sys.exit(1 if bad else 0)
