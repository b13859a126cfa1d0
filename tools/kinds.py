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
    rc, out, blocks = oracle.decode(srcs[0], cap=size, want_trace=True)
    b = blocks[0]
    print("%-8s ok=%s kernel %.3f ms  ratio %.2f  block0: type %d lit_type %d nlit %d nseq %d" % (
        kind, ok, mzd.last_kernel_ms(0), size * n / cp.comp_sizes.sum(), b["block_type"], b["lit_type"], b["n_lit"], b["n_seq"]), flush=True)
