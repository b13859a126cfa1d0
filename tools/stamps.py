"""Diagnostic: per-phase cycles of one workgroup (libmzd_diag.so, -DMZD_STAMPS).  Not a benchmark."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), os.environ.get("MZD_DIAG_SO", "libmzd_diag.so"))
import fuse_zstd_amd as mzd, corpus
mzd.init()
kind = sys.argv[1] if len(sys.argv) > 1 else "json"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cp = corpus.build_corpus(kind, 2, [size] * n)
srcs = [cp.comp_file(i).tobytes() for i in range(n)]
for rep in range(2):
    res = mzd.decode_batch(srcs, [size] * n)
assert all(st == 0 for st, _ in res)
st = (C.c_uint64 * 24)()
api.lib().mzd_debug_stamps(0, st)
names = ["hdr", "K1 weights+parse", "(count) walker slow-window iterations", "K2 literals+seqhdr", "K3 tables", "K4 seq decode", "K5 execute", "K7 xxh64"]
tot = sum(st[:8])
print("kernel ms", mzd.last_kernel_ms(0), "files", n)
for nm, v in zip(names, st[:8]):
    print("%-22s %10d cycles %5.1f%%" % (nm, v, 100.0 * v / max(tot, 1)))
cn = ["wait plan", "loop top", "chunk setup + classify + issue HBM loads", "lit/old-match regs -> LDS (waits for the loads)", "matches from previous runs (LDS) / big (HBM)", "rounds LDS->LDS", "fence before flush", "flush"]
print("cycles after block start: tables ready %d, literals ready %d, walker done %d, planner done %d, copier done %d, hasher done %d" % (st[21], st[20], st[16], st[19], st[17], st[18]))
if os.environ.get("MZD_WALKSTAT"):  # (a build with -DMZD_STAMPS -DMZD_EXP_WALKSTAT: the walker's statistics in the copier's slots)
    w = list(st[8:16])
    print("walker: ring refills %d (%d cycles), assembly runs %d (%d cycles), void groups taken again %d (%d cycles), between %d cycles" % (w[1], w[0], w[3], w[2], w[5], w[4], w[6]))
    sys.exit(0)
print("copier wavefront:")
for nm, v in zip(cn, st[8:]):
    print("   %-26s %10d cycles" % (nm, v))

import numpy as np
buf = (C.c_uint64 * (12 * 2048))()
ns = api.lib().mzd_debug_tfin_all(0, buf, 2048)
if ns > 0:
    arr = np.frombuffer(buf, dtype=np.uint64)[: ns * 12].reshape(ns, 12).astype(np.float64)
    arr = arr[arr[:, 0] > 0][: n]
    if len(arr):
        names6 = ["walker done", "copier done", "hasher done", "planner done", "first literal stream done (copier wave)", "tables ready", "headers parsed", "Huffman weights decoded", "Huffman table filled", "copier started", "task start -> block start", "task start -> task end"]
        print("over %d workgroups (cycles after block start): " % len(arr) + "; ".join(("%s mean %.1f max %.0f" % (nm, arr[:, k].mean(), arr[:, k].max())) if nm.startswith("(count)") else ("%s mean %.0fK max %.0fK" % (nm, arr[:, k].mean() / 1e3, arr[:, k].max() / 1e3)) for k, nm in enumerate(names6)))
