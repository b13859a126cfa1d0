"""Stress (not a benchmark): a few multi-block shapes, each decoded MANY times under several drivers, every output compared --
rare races between block tasks (about once in a hundred runs) only show under repetition of the same input.
  python tools/stress_repeat.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd as mzd, corpus
mzd.init()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
shapes = [("json", [1 << 20] * 40, 3), ("text", [3 << 20] * 12, 3), ("int32", [300000] * 200, 1), ("xray", [2 << 20] * 16, 3), ("repeats", [1 << 20] * 50, 9),
          ("json", [200000, 5000, 1 << 20, 131073, 70000, 2 << 20] * 20, 3), ("markup", [140000] * 300, 3)]
bad_total = 0
for kind, sizes, level in shapes:
    cp = corpus.build_corpus(kind, 71, sizes, level=level)
    srcs = [cp.comp_file(i).tobytes() for i in range(len(sizes))]
    want = [cp.raw_file(i).tobytes() for i in range(len(sizes))]
    for drv in (0, 4, 5, 1):
        mzd.set_driver(drv)
        bad = 0
        for rep in range(reps):
            res = mzd.decode_batch(srcs, sizes)
            if any(st != 0 or out != want[i] for i, (st, out) in enumerate(res)):
                bad += 1
        print("%-8s %3d files of %7d.. level %d driver %d: %d bad of %d" % (kind, len(sizes), sizes[0], level, drv, bad, reps), flush=True)
        bad_total += bad
mzd.set_driver(0)
print("stress_repeat:", "ok" if not bad_total else "BAD %d" % bad_total)
sys.exit(1 if bad_total else 0)
