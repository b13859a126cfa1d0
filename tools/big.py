"""Diagnostic: kernel time of n files of one size under both drivers (mzd_debug_set_driver 1: a workgroup per file, 5: block tasks, 4: block tasks with blocks resolved ahead)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd as mzd, corpus
mzd.init()
kind = sys.argv[1] if len(sys.argv) > 1 else "json"
for size, n in ((1 << 20, 1), (1 << 20, 8), (1 << 20, 64), (1 << 18, 64), (1 << 20, 400)):
    cp = corpus.build_corpus(kind, 5, [size] * n)
    srcs = [cp.comp_file(i).tobytes() for i in range(n)]
    line = "%s %4d x %7d B:" % (kind, n, size)
    for drv in ("1", "5", "4"):
        mzd.set_driver(int(drv))
        for rep in range(3):
            res = mzd.decode_batch(srcs, [size] * n)
        ok = all(st == 0 and out == cp.raw_file(i).tobytes() for i, (st, out) in enumerate(res))
        line += "  driver %s %.3f ms (%s)" % (drv, mzd.last_kernel_ms(0), "ok" if ok else "BAD")
    print(line, flush=True)
