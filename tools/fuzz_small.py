"""Diagnostic: mutated SMALL frames through the LDS small-file kernel (mzd_debug_set_driver 3): every data class x three levels x
five sizes up to 8 KiB, 1..3 random byte flips or a truncation each, with and without a dictionary, decoded in one launch and
compared with the oracle (status, and bytes where both accept).  Exits non-zero on any mismatch.  python tools/fuzz_small.py [seed]
tests/test_gpu_parity.py holds the committed, exhaustive single-byte form for six frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corpus, oracle, fuse_zstd_amd as mzd
mzd.init()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
rng = np.random.RandomState(seed)
cases = []  # (comp, cap, dict bytes or None, handle)
for kind in ("json", "text", "markup", "int32", "dna", "xray", "random", "repeats"):
    for level in (1, 3, 19):
        sizes = [300, 1000, 2500, 4096, 8000]
        # (every other corpus without the content checksum: there the kernel's own checks decide, not XXH64)
        cp = corpus.build_corpus(kind, 100 + seed, sizes, level=level, checksum=(level != 3))
        for i, size in enumerate(sizes):
            good = cp.comp_file(i).tobytes()
            cases.append((good, size, None, 0))
            for _ in range(30):
                b = bytearray(good)
                for _ in range(int(rng.randint(1, 4))):
                    b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
                cases.append((bytes(b), size, None, 0))
            for _ in range(6):
                cases.append((good[:int(rng.randint(0, len(good)))], size, None, 0))
            cases.append((good, size - 1, None, 0))   # a destination one byte short
sizes = [int(x) for x in np.random.RandomState(55).randint(300, 3001, size=300)]
d = corpus.train_dict("json", 5, sizes[:200], cap=40000)
h = mzd.load_dict(d)
cp = corpus.build_corpus("json", 5, sizes, dictionary=d)
for i in range(300):
    good = cp.comp_file(i).tobytes()
    cases.append((good, sizes[i], d, h))
    for _ in range(8):
        b = bytearray(good)
        for _ in range(int(rng.randint(1, 3))):
            b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
        cases.append((bytes(b), sizes[i], d, h))
    cases.append((good[:int(rng.randint(0, len(good)))], sizes[i], d, h))
mzd.set_driver(3)
res = mzd.decode_batch([c for c, _, _, _ in cases], [cap for _, cap, _, _ in cases], [hh for _, _, _, hh in cases])
mzd.set_driver(0)
c = mzd.debug_counters(0)
bad = 0
for i, ((comp, cap, dd, _), (st, out)) in enumerate(zip(cases, res)):
    rc, want = oracle.decode(comp, cap=cap, dictionary=dd)
    if st != rc or (st == 0 and out != want):
        bad += 1
        if bad < 10: print("MISMATCH case", i, "gpu", st, "oracle", rc, "len", len(comp), "cap", cap, "dict", dd is not None)
print("seed", seed, "cases", len(cases), "bad", bad, "accepted", sum(1 for st, _ in res if st == 0))
sys.exit(1 if bad else 0)
