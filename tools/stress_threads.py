"""Stress of the host path (not a benchmark): several threads call mzd_decode_batch at once on batches of very different shape --
thousands of small files, a few very big ones (the whole-device path), mixes; pinned and pageable buffers -- and every output is
compared.  Lanes, stagings, the copy pool and the whole-device guard are shared between the calls."""
import os, sys, threading, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuse_zstd_amd as mzd, corpus
from fuse_zstd_amd import api
mzd.init()
L = api.lib()
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
shapes = [("json", [4096] * 3000), ("json", [1 << 20] * 6), ("text", [16 << 20]), ("json", [131072] * 300), ("xray", [300000] * 20),
          ("markup", [int(x) for x in np.exp(np.random.RandomState(3).uniform(np.log(1000), np.log(1 << 20), size=400))]), ("repeats", [5 << 20] * 2)]
corp = [corpus.build_corpus(k, 70 + i, s) for i, (k, s) in enumerate(shapes)]
bad = []
def worker(tid):
    rng = random.Random(100 + tid)
    for r in range(rounds):
        ci = rng.randrange(len(corp)); cp = corp[ci]
        pinned = rng.random() < 0.5
        end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
        if pinned:
            hin = mzd.HostBuffer(len(cp.comp)); hin.a[:] = cp.comp; src = hin.a
            hout = mzd.HostBuffer(end + 64); out = hout.a; out[:end] = 0
        else:
            src = cp.comp; out = np.zeros(end + 64, dtype=np.uint8)
        jobs = api.make_jobs([src.ctypes.data + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.ctypes.data + int(o) for o in cp.raw_offs], cp.raw_sizes)
        rc = L.mzd_decode_batch(jobs, cp.nfiles)
        ok = rc == 0 and all(j.status == 0 for j in jobs) and bool((out[:end] == cp.raw[:end]).all())
        if not ok:
            bad.append((tid, r, ci, rc))
        print("thread %d round %d: shape %d (%s, %d files, %s): %s" % (tid, r, ci, shapes[ci][0], cp.nfiles, "pinned" if pinned else "pageable", "ok" if ok else "BAD"), flush=True)
th = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
for t in th: t.start()
for t in th: t.join()
print("stress", "ok" if not bad else "BAD %r" % bad[:5], nthreads * rounds, "calls")
sys.exit(1 if bad else 0)
