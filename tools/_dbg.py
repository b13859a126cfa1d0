import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import corpus, oracle, fuse_zstd_amd as mzd
mzd.init()
rng = np.random.RandomState(77)
cases = []
for kind, size in (("json", 600000), ("text", 400000), ("xray", 300000), ("repeats", 500000)):
    cp = corpus.build_corpus(kind, 21, [size])
    good = cp.comp_file(0).tobytes()
    for _ in range(120):
        b = bytearray(good); pos = int(rng.randint(0, len(b))); b[pos] ^= int(rng.randint(1, 256)); cases.append((bytes(b), size, pos))
    break
comp, cap, pos = cases[111]
print("case 111: mutated byte", pos, "of", len(comp), "oracle", oracle.decode(comp, cap=cap)[0], "unpinned", oracle.last_verdict_unpinned())
rc, out, blocks = oracle.decode(bytes(cases[111][0]), cap=cap, want_trace=True)
print([ (b["block_type"], b["n_seq"], b["regen"]) for b in blocks])
for drv in (0, 1, 2, 4, 5):
    mzd.set_driver(drv)
    print("driver", drv, "alone:", [mzd.decode(comp, cap)[0] for _ in range(3)], mzd.last_kernel_name(0))
mzd.set_driver(0)
res = mzd.decode_batch([c for c, _, _ in cases], [cap for _, cap, _ in cases])
print("batch statuses differing:", [(i, st, oracle.decode(c, cap=cp_)[0]) for i, ((c, cp_, _), (st, _)) in enumerate(zip(cases, res)) if st != oracle.decode(c, cap=cp_)[0]])
