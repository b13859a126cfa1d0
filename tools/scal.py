import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd as mzd, corpus, oracle
mzd.init()
kind = sys.argv[1]
cp = corpus.build_corpus(kind, 3, [131072] * 16)
for i in range(8):
    src = cp.comp_file(i).tobytes()
    for rep in range(3):
        res = mzd.decode_batch([src], [131072])
    rc, out, blocks = oracle.decode(src, cap=131072, want_trace=True)
    b = blocks[0]
    print(kind, "file", i, "alone: kernel ms %.3f" % mzd.last_kernel_ms(0), "comp", len(src), "streams", b["lit_streams"], "huf_max_bits", b["huf_max_bits"], "nlit", b["n_lit"], "nseq", b["n_seq"], flush=True)
