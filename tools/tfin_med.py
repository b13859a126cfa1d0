"""Diagnostic: medians of the per-workgroup role finish times (libmzd_tfin.so / MZD_DIAG_SO).  Not a benchmark."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as api
api._SO = os.path.join(os.path.dirname(api._SO), os.environ.get("MZD_DIAG_SO", "libmzd_tfin.so"))
import fuse_zstd_amd as mzd, corpus
import numpy as np
mzd.init()
kind, size, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if kind == "cfg5":  # the shared-dictionary shape of bench.py --workload cfg5
    sizes = [int(x) for x in np.random.RandomState(55).randint(300, 3001, size=n)]
    d = corpus.train_dict("json", 5, [int(x) for x in np.random.RandomState(55).randint(300, 3001, size=4000)], cap=112640)
    did = mzd.load_dict(d)
    cp = corpus.build_corpus("json", 5, sizes, dictionary=d)
    srcs = [cp.comp_file(i).tobytes() for i in range(n)]
    for rep in range(2):
        res = mzd.decode_batch(srcs, sizes, dict_ids=[did] * n)
else:
    cp = corpus.build_corpus(kind, 2, [size] * n)
    import torch  # device-resident: ONE launch on the whole device (the host path would cut the batch into chunks)
    dcomp = torch.from_numpy(cp.comp).cuda()
    dout = torch.zeros(int(cp.raw_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=torch.uint8, device="cuda")
    jobs = api.make_jobs([dcomp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [dout.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
    for rep in range(3):
        res = api.decode_batch_device(0, jobs)
        torch.cuda.synchronize()
assert all(st == 0 for st, _ in res)
buf = (C.c_uint64 * (12 * 2048))()
ns = api.lib().mzd_debug_tfin_all(0, buf, 2048)
arr = np.frombuffer(buf, dtype=np.uint64)[: ns * 12].reshape(ns, 12).astype(np.float64)
arr = arr[arr[:, 0] > 0]
names = ["walker", "copier", "hasher", "planner", "lit stream 1", "tables", "headers", "huf weights", "huf table", "copier start", "file start -> block start", "file start -> file end"]
print("kernel ms", mzd.last_kernel_ms(0), "wgs", len(arr))
if os.environ.get("MZD_TFIN_ABS"):  # (libmzd_exp.so built with -DMZD_TFIN_ABS: columns 10 / 11 are absolute clock values)
    st, en = arr[:, 10], arr[:, 11]
    print("100 MHz clock, microseconds.  starts: spread %.1f (p50 - min %.1f, p99 - min %.1f); ends - first start: min %.1f, p50 %.1f, max %.1f" % ((st.max() - st.min()) / 1e2, (np.median(st) - st.min()) / 1e2, (np.percentile(st, 99) - st.min()) / 1e2, (en.min() - st.min()) / 1e2, (np.median(en) - st.min()) / 1e2, (en.max() - st.min()) / 1e2))
    sys.exit(0)
print("median: " + "; ".join("%s %.0fK" % (nm, np.median(arr[:, k]) / 1e3) for k, nm in enumerate(names)))
print("max:    " + "; ".join("%s %.0fK" % (nm, np.max(arr[:, k]) / 1e3) for k, nm in enumerate(names)))
