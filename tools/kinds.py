"""Diagnostic: kernel time per corpus kind (N files of 128 KiB each, product library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuse_zstd_amd.api as _api
if os.environ.get('MZD_AB_SO'): _api._SO = os.path.join(os.path.dirname(_api._SO), os.environ['MZD_AB_SO'])
import fuse_zstd_amd as mzd, corpus, oracle
mzd.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
size = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
for kind in corpus.KINDS:
    cp = corpus.build_corpus(kind, 3, [size] * n)
    srcs = [cp.comp_file(i).tobytes() for i in range(n)]
    import torch  # device-resident (one launch on the whole device; the host path would split the batch into chunks)
    dcomp = torch.from_numpy(cp.comp).cuda()
    dout = torch.zeros(int(cp.raw_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=torch.uint8, device="cuda")
    jobs = mzd.api.make_jobs([dcomp.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes, [dout.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes)
    for rep in range(3):
        mzd.api.decode_batch_device(0, jobs)
        torch.cuda.synchronize()
    got = dout.cpu().numpy()
    ok = all(j.status == 0 for j in jobs) and all(bytes(got[int(cp.raw_offs[i]):int(cp.raw_offs[i]) + size]) == cp.raw_file(i).tobytes() for i in range(0, n, 17))
    # the first block of 24 of the files (ONE file's says little: an `xray` file has anything from 0 to 4 500 sequences)
    import numpy as np
    bs = [oracle.decode(srcs[i], cap=size, want_trace=True)[2][0] for i in range(0, n, max(1, n // 24))]
    nl = np.array([b["n_lit"] for b in bs]); ns = np.array([b["n_seq"] for b in bs])
    print("%-8s ok=%s kernel %.3f ms  ratio %.2f  first blocks: type %d lit_type %d; literals min %d median %d max %d; sequences min %d median %d max %d" % (
        kind, ok, mzd.last_kernel_ms(0), size * n / cp.comp_sizes.sum(), bs[0]["block_type"], bs[0]["lit_type"], nl.min(), np.median(nl), nl.max(), ns.min(), np.median(ns), ns.max()), flush=True)
