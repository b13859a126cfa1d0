"""Diagnostic: when every workgroup (one wavefront) of the small-file kernel starts and ends, and where it runs, from a library built with
`make sstamps`: the launch's fill and tail, per XCD / CU / SIMD.   python tools/lds_wg.py [cfg4] [files]   (G / XG in the environment as for lds_stamps.py)"""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), os.environ.get("MZD_SO", "libmzd_sstamps.so"))
import bench, corpus, fuse_zstd_amd as mzd
mzd.init()
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_FILES[wl]
kind, cfg, km, _ = bench.WORKLOADS[wl]
cp = corpus.build_corpus(kind, cfg, bench.file_sizes(wl, n, 0, 1), kind_mod=km)
import torch
dev = torch.device("cuda:0")
comp = torch.from_numpy(cp.comp).to(dev)
out = torch.zeros(int(cp.raw_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=torch.uint8, device=dev)
jobs = api.make_jobs([comp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [out.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes, None)
mzd.set_driver(3)
if os.environ.get('G'): api.lib().mzd_debug_host_path(0, 4, int(os.environ['G']))
if os.environ.get('XG'): api.lib().mzd_debug_host_path(0, 5, int(os.environ['XG']))
NW = 3072
for rep in range(3):
    res = mzd.decode_batch_device(0, jobs)
    assert all(st == 0 for st, _ in res)
    st = (C.c_uint64 * (16 * NW))()
    api.lib().mzd_debug_small_wg_stamps.argtypes = [C.c_int, C.c_void_p, C.c_int]
    api.lib().mzd_debug_small_wg_stamps(0, st, NW)
a = np.array(list(st), dtype=np.uint64).reshape(NW, 16)
a = a[a[:, 1] != 0]
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64); base = t0.min()
t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0  # microseconds
hw = a[:, 2] & np.uint64(0xFFFFFFFF); xcc = (a[:, 2] >> np.uint64(32)) & np.uint64(0xF)
simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(15); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
groups = a[:, 3] & np.uint64(0xFFFF); rounds = ((a[:, 3] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64); steps = ((a[:, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
tent = ((a[:, 3] >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64) / 100.0  # entropy phases of the first group, microseconds
print("%s %s: kernel %.3f ms, %d workgroups; groups per workgroup %s" % (wl, os.environ.get("G", "auto") + "/" + os.environ.get("XG", "-"), mzd.last_kernel_ms(0), len(a), dict(collections.Counter(groups.tolist()))))
q = lambda v: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max())
print("  start (us after the first):", q(t0)); print("  end:                       ", q(t1)); print("  lifetime:                  ", q(t1 - t0))
cuid = (xcc.astype(np.int64) * 8 + se.astype(np.int64)) * 32 + sh.astype(np.int64) * 16 + cu.astype(np.int64)
per_cu = collections.Counter(cuid.tolist()); print("  CUs used %d; workgroups per CU: %s" % (len(per_cu), dict(collections.Counter(per_cu.values()))))
per_simd = collections.Counter((cuid * 4 + simd.astype(np.int64)).tolist()); print("  wavefronts per SIMD: %s" % dict(collections.Counter(per_simd.values())))
for k in sorted(set(per_simd.values())):
    sel = np.array([per_simd[int(c) * 4 + int(s)] == k for c, s in zip(cuid, simd)])
    print("    SIMDs holding %d: lifetime median %.1f  max %.1f us; end median %.1f max %.1f" % (k, np.median((t1 - t0)[sel]), (t1 - t0)[sel].max(), np.median(t1[sel]), t1[sel].max()))
life = t1 - t0
print("  entropy phases (first group):", q(tent)); print("  execution + rest:           ", q(life - tent))
print("  match rounds per workgroup:", q(rounds.astype(float)), " steps:", q(steps.astype(float)))
print("  correlation of lifetime with rounds %.2f, with steps %.2f, of execution time with rounds %.2f" % (np.corrcoef(life, rounds)[0, 1], np.corrcoef(life, steps)[0, 1], np.corrcoef(life - tent, rounds)[0, 1]))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("    XCC %d: %4d workgroups, lifetime median %.1f max %.1f, entropy median %.1f" % (x, m.sum(), np.median(life[m]), life[m].max(), np.median(tent[m])))
ncu = np.array([per_cu[int(c)] for c in cuid])
for k in sorted(set(ncu.tolist())):
    m = ncu == k
    print("    CUs holding %d: lifetime median %.1f max %.1f, entropy median %.1f" % (k, np.median(life[m]), life[m].max(), np.median(tent[m])))
ph = a[:, 4:14].astype(np.int64)  # clocks at the phase boundaries 0..9 of the first group (8: XXH64 of the last pass, 9: end of the group)
names = ["input -> LDS", "headers", "Huffman weights + table", "Huffman streams", "sequence header", "FSE tables", "walk + extract", "execution passes (to the last XXH64)", "flush + results"]
for k in range(9):
    d = (ph[:, k + 1] - ph[:, k]) / 100.0
    print("    %-40s %s" % (names[k], q(d)))
slow = tent > np.percentile(tent, 90)
print("  the slowest tenth by entropy time: SIMD occupancy %s, per CU %s" % (dict(collections.Counter(per_simd[int(c) * 4 + int(s)] for c, s in zip(cuid[slow], simd[slow]))), dict(collections.Counter(ncu[slow].tolist()))))
print("  histogram of entropy time (10 us bins from 100):", np.histogram(tent, bins=np.arange(100, 230, 10))[0].tolist())
late = t0 > np.percentile(t0, 50) + 20
print("  workgroups that started > 20 us after the median start: %d; their lifetime median %.1f" % (late.sum(), np.median((t1 - t0)[late]) if late.any() else 0))
