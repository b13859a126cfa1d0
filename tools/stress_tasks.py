"""Stress of the block-task driver (not a benchmark): many mixed batches in one process, every output compared.
Races between tasks (hand-over of tables / repeat offsets / positions / checksum state) would show up as mismatches."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd as mzd, corpus
mzd.init()
mzd.set_driver(int(os.environ.get("MZD_DRIVER", "2")))
rng = random.Random(12345)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
total = 0
for r in range(rounds):
    kind = rng.choice(["json", "text", "markup", "int32", "dna", "xray", "repeats", "random"])
    n = rng.choice([1, 3, 17, 120, 700])
    sizes = [rng.choice([1, 100, 4096, 70000, 131072, 131073, 262144, 400000, 1 << 20, 3 << 20]) if rng.random() < 0.3 else rng.randint(1, 300000) for _ in range(n)]
    if sum(sizes) > (96 << 20):
        sizes = sizes[:40]
    level = rng.choice([1, 3, 3, 9, 19])
    cp = corpus.build_corpus(kind, 11 + r, sizes, level=level)
    res = mzd.decode_batch([cp.comp_file(i).tobytes() for i in range(len(sizes))], sizes)
    bad = [(i, sizes[i], st) for i, (st, out) in enumerate(res) if st != 0 or out != cp.raw_file(i).tobytes()]
    total += len(sizes)
    print("round %d: %s level %d, %d files, %d MiB: %s" % (r, kind, level, len(sizes), sum(sizes) >> 20, "ok" if not bad else "BAD %r" % bad[:5]), flush=True)
    if bad:
        sys.exit(1)
print("stress ok:", total, "files")
