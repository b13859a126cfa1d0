"""Diagnostic: kernel time per corpus kind (N files of 128 KiB each, product library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd as mzd, corpus, oracle
mzd.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
size = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
for kind in corpus.KINDS:
    cp = corpus.build_corpus(kind, 3, [size] * n)
    srcs = [cp.comp_file(i).tobytes() for i in range(n)]
    for rep in range(2):
        res = mzd.decode_batch(srcs, [size] * n)
    ok = all(st == 0 and out == cp.raw_file(i).tobytes() for i, (st, out) in enumerate(res))
    rc, out, blocks = oracle.decode(srcs[0], cap=size, want_trace=True)
    b = blocks[0]
    print("%-8s ok=%s kernel %.3f ms  ratio %.2f  block0: type %d lit_type %d nlit %d nseq %d" % (
        kind, ok, mzd.last_kernel_ms(0), size * n / cp.comp_sizes.sum(), b["block_type"], b["lit_type"], b["n_lit"], b["n_seq"]), flush=True)
