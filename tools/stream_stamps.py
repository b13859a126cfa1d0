"""Diagnostic (libmzd_exp.so built with EXPFLAGS=-DMZD_EXP_STREAMSTAMP): which wavefront takes which Huffman stream of a literal-heavy block when, and when
it is done, per workgroup.   python tools/stream_stamps.py xray 131072 1000"""
import ctypes as C, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), os.environ.get("MZD_DIAG_SO", "libmzd_exp.so"))
import fuse_zstd_amd as mzd, corpus
import numpy as np, torch, oracle
mzd.init()
kind, size, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cp = corpus.build_corpus(kind, 2, [size] * n)
dcomp = torch.from_numpy(cp.comp).cuda()
dout = torch.zeros(int(cp.raw_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=torch.uint8, device="cuda")
jobs = api.make_jobs([dcomp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [dout.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
for rep in range(3):
    res = api.decode_batch_device(0, jobs); torch.cuda.synchronize()
if not os.environ.get("MZD_NOASSERT"): assert all(st == 0 for st, _ in res)
buf = (C.c_uint64 * (12 * 2048))()
ns = api.lib().mzd_debug_tfin_all(0, buf, 2048)
a = np.frombuffer(buf, dtype=np.uint64)[: ns * 12].reshape(ns, 12)
a = a[a[:, 8] > 0]
take = (a[:, 0:4] >> np.uint64(2)).astype(np.float64) / 1e3; who = (a[:, 0:4] & np.uint64(3)).astype(int)
hi = lambda c: (a[:, c] >> np.uint64(32)).astype(np.float64) / 1e3
lo = lambda c: (a[:, c] & np.uint64(0xFFFFFFFF)).astype(np.float64) / 1e3
print("kernel ms %.3f, %d workgroups; Huffman table filled at (median) %.0fK" % (mzd.last_kernel_ms(0), len(a), np.median(a[:, 8]) / 1e3))
for st in range(4):
    print("  stream %d: taken at median %.0fK (p10 %.0fK, p90 %.0fK); by wavefront %s" % (st, np.median(take[:, st]), np.percentile(take[:, st], 10), np.percentile(take[:, st], 90), dict(collections.Counter(who[:, st].tolist()))))
q = lambda v: "median %.0fK (p10 %.0fK, p90 %.0fK)" % (np.median(v), np.percentile(v, 10), np.percentile(v, 90))
print("  wavefront 0: role entry", q(hi(4)), "| past its first get_seq", q(lo(4)), "| helper entry", q(hi(5)), "| first stream", q(lo(5)))
print("  wavefront 3: role entry", q(hi(6)), "| past its first get_seq", q(lo(6)), "| helper entry", q(hi(7)), "| first stream", q(lo(7)))
per_wave = collections.Counter(tuple(sorted(collections.Counter(w.tolist()).values(), reverse=True)) for w in who)
print("  streams per wavefront (sorted counts) across workgroups:", dict(per_wave))
late = take.max(axis=1) - take.min(axis=1)
print("  last take - first take: median %.0fK, p90 %.0fK, max %.0fK" % (np.median(late), np.percentile(late, 90), late.max()))

# what the helpers were doing meanwhile: their own roles.  Sequences per block over the same files (oracle trace), sorted, against
# the helper wavefronts' arrival at the stream queue, sorted: the block with k sequences is walked and planned first.
nseq = np.sort(np.array([oracle.decode(cp.comp_file(i).tobytes(), cap=size, want_trace=True)[2][0]["n_seq"] for i in range(n)]))
ent = np.sort(hi(5))
print("  sequences in the files' (first) blocks: min %d, p10 %d, median %d, p90 %d, max %d; blocks without any: %d of %d" % (nseq.min(), np.percentile(nseq, 10), np.median(nseq), np.percentile(nseq, 90), nseq.max(), (nseq == 0).sum(), n))
if len(ent) == n and nseq.max() > 0:
    k = np.polyfit(nseq, ent, 1)
    print("  wavefront 0 arrives at the stream queue at ~ %.0fK + %.0f cycles per sequence (sorted against sorted; correlation %.3f)" % (k[1], k[0] * 1e3, np.corrcoef(nseq, ent)[0, 1]))
