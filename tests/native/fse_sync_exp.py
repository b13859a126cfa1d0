"""Experiment: self-synchronisation of zstd's sequence bitstream (see fse_sync_exp.c).  python tests/native/fse_sync_exp.py [lanes]"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import corpus
so = os.path.join(HERE, "fse_sync_exp.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "fse_sync_exp.c")])
L = C.CDLL(so)
L.ozs_decode.restype = C.c_int
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
KINDS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["json", "text", "markup", "int32", "dna", "xray", "repeats"]
for kind in KINDS:
    for size in (131072,):
        try:
            cp = corpus.build_corpus(kind, 2, [size] * 24)
        except Exception as e:
            print(kind, "skipped:", e); continue
        L.exp_reset(lanes)
        for i in range(24):
            src = cp.comp_file(i).tobytes()
            dst = C.create_string_buffer(size + 64)
            out = C.c_size_t(0)
            rc = L.ozs_decode(src, C.c_size_t(len(src)), dst, C.c_size_t(size), C.byref(out), None, C.c_size_t(0), None)
            assert rc == 0, rc
        hist = (C.c_uint64 * 4096)(); misc = (C.c_uint64 * 8)(); rounds = (C.c_uint64 * 258)()
        L.exp_get(hist, misc, rounds)
        h = np.array(hist[:], dtype=np.float64); never, starts, blocks, seqs, bits, st_tot, st_ser = [int(x) for x in misc[:7]]
        if blocks == 0:
            print("%-6s no block with >= 256 sequences" % kind); continue
        c = np.cumsum(h); tot = c[-1] + never
        q = lambda f: int(np.searchsorted(c, f * tot)) if c[-1] >= f * tot else -1
        r = np.array(rounds[:]); rr = [(i, int(v)) for i, v in enumerate(r) if v]
        print("%-6s blocks %3d seq/block %6d bits/seq %5.1f | slice %4d seq | sync after (sequences): p50 %4d p90 %4d p99 %4d max %4d never %d/%d | rounds %s | parallel steps/block %.0f vs serial %.0f (x%.1f)" % (
            kind, blocks, seqs // blocks, bits / seqs, seqs // blocks // lanes, q(0.5), q(0.9), q(0.99), int(np.nonzero(h)[0].max()) if h.any() else -1, never, starts, rr, st_tot / blocks, st_ser / blocks, st_ser / max(st_tot, 1)))
