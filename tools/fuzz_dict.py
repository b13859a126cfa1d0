"""Diagnostic: mutated dictionary frames (config-5 shape), multi-frame files with skippable frames, and truncations of
config-2-sized frames, decoded on the GPU in one launch and compared with the oracle; exits non-zero on any mismatch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corpus, oracle, fuse_zstd_amd as mzd
mzd.init()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.RandomState(seed)
Z = oracle.LibZstd
assert Z.available()
cases = []  # (comp, cap, dict bytes or None, dict handle)
# 1. dictionary frames
sizes = [int(x) for x in np.random.RandomState(55).randint(300, 3001, size=400)]
d = corpus.train_dict("json", 5, sizes[:300], cap=40000)
h = mzd.load_dict(d)
cp = corpus.build_corpus("json", 5, sizes, dictionary=d)
for i in range(0, 400, 2):
    good = cp.comp_file(i).tobytes()
    for _ in range(6):
        b = bytearray(good)
        for _ in range(int(rng.randint(1, 3))):
            b[int(rng.randint(0, len(b)))] ^= int(rng.randint(1, 256))
        cases.append((bytes(b), sizes[i], d, h))
    cases.append((good, sizes[i], None, 0))           # the dictionary is missing
    cases.append((good[:int(rng.randint(0, len(good)))], sizes[i], d, h))
# 2. multi-frame files: frame + skippable + frame, mutated
for k in range(120):
    a = Z.compress(corpus.gen("json", 60 + k, 1, 3000), 3, True)
    b2 = Z.compress(corpus.gen("text", 61 + k, 1, 2000), 3, bool(k & 1))
    skip = (0x184D2A50 + (k & 15)).to_bytes(4, "little") + (7).to_bytes(4, "little") + b"skipped"
    f = bytearray(a + skip + b2)
    if k % 3:
        for _ in range(int(rng.randint(1, 3))):
            f[int(rng.randint(0, len(f)))] ^= int(rng.randint(1, 256))
    cases.append((bytes(f), 5000 if k % 5 else 4000, None, 0))
# 3. truncations of 128 KiB frames
for kind, sd in (("json", 71), ("text", 72)):
    good = corpus.build_corpus(kind, sd, [131072]).comp_file(0).tobytes()
    for _ in range(150):
        cases.append((good[:int(rng.randint(0, len(good)))], 131072, None, 0))
res = mzd.decode_batch([c[0] for c in cases], [c[1] for c in cases], [c[3] for c in cases])
bad = 0
for i, ((comp, cap, dd, _), (st, out)) in enumerate(zip(cases, res)):
    rc, want = oracle.decode(comp, cap=cap, dictionary=dd)
    if st != rc or (st == 0 and out != want):
        bad += 1
        if bad <= 12:
            print("MISMATCH case", i, "gpu", st, "oracle", rc, "dict" if dd else "", len(comp))
print("cases", len(cases), "bad", bad, "accepted", sum(1 for st, _ in res if st == 0))
sys.exit(1 if bad else 0)
