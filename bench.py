#!/usr/bin/env python3
"""bench.py -- decompressed GiB/s of the zstd decode path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (the whole-file decode fuse-zstd runs on open(), reference
src/main.rs:463-467) over one batch of synthetic .zst files already resident in HBM: the timed
region contains only kernel launches (device-resident compressed bytes in, decompressed bytes
left in HBM).  Workload at N=1 = BASELINE.json configs[1]: 1 000 independent 128 KiB
single-block JSON frames written like the reference's writer (level 3, checksum, pledged size;
src/main.rs:781-791).  With N GPUs every rank takes files r, r+N, r+2N, ... of an N x 1000-file
corpus (file i -> GPU i mod N, no collective; weak scaling).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg4lu|cfg5] [--files F]

The other workloads are BASELINE's remaining configurations (parity cases with a selectable bench line): cfg3 the
Silesia-proxy mix, cfg4 10 000 x 4 KiB files, cfg4lu 10 000 files log-uniform 4 KiB..1 MiB (multi-block frames:
the block-task driver), cfg5 50 000 small records with one shared dictionary.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      HBM bound.  achieved = algorithmic bytes per launch (sum of compressed bytes read once
                + decompressed bytes written once) / average kernel duration measured
                here with events on the launch stream.  traffic = HBM bytes per launch from the PMC
                passes recorded in profiles/ (null until such a pass exists for this workload).
  cpu_baseline  rank 0 at N=1: the reference's CPU codec (system libzstd through dlopen) timed on
                this box's host cores on the same files, repeated to ~10 s of CPU work.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured copy
GIB = float(1 << 30)

WORKLOADS = {
    # name: (corpus kind, cfg id, kind_mod, sizes(nfiles, rank, world) -> list, default files per GPU, description)
    "cfg2": ("json", 2, 0, "1000 x 128 KiB single-block JSON frames, zstd level 3, checksum+FCS (BASELINE configs[1])"),
    "cfg3": ("text", 3, 7, "Silesia-proxy mix as 128 KiB single-block frames, level 3 (BASELINE configs[2]; Silesia itself is not on the box)"),
    "cfg4": ("json", 4, 0, "4 KiB JSON files, parallel-files.fio shape (BASELINE configs[3])"),
    "cfg4lu": ("json", 4, 0, "JSON files log-uniform 4 KiB..1 MiB (BASELINE configs[3] variant)"),
    "cfg5": ("json", 5, 0, "JSON records of 300..3000 B, one shared ZDICT-trained dictionary (BASELINE configs[4]; 50 000 files per GPU by default)"),
}


def file_sizes(workload, nfiles, rank, world):
    if workload in ("cfg2", "cfg3"):
        return [131072] * nfiles
    if workload == "cfg4":
        return [4096] * nfiles
    if workload == "cfg5":
        rng = np.random.RandomState(55)
        allsz = rng.randint(300, 3001, size=nfiles * world)
        return [int(x) for x in allsz[rank::world]]
    rng = np.random.RandomState(1234)  # same table on every rank; rank r takes entries r, r+world, ...
    allsz = np.exp(rng.uniform(np.log(4096), np.log(1 << 20), size=nfiles * world)).astype(np.int64)
    return [int(x) for x in allsz[rank::world]]


def cpu_baseline(cp, budget_s=10.0):
    """The reference's CPU path on this host: libzstd (dlopen).  B2 = all cores, one file per task,
    one-shot decode with a reused DCtx per thread; B1 = one thread, streaming through an 8 KiB
    buffer (the shape copy_decode + io::copy produce, reference src/main.rs:463)."""
    import ctypes as C
    import oracle
    cores = os.cpu_count() or 1
    nfiles = cp.nfiles
    total_out = int(cp.raw_sizes.sum())
    if oracle.LibZstd.available():
        L = oracle.lib()
        offs = np.ascontiguousarray(cp.comp_offs, dtype=np.uint64)
        sizes = np.ascontiguousarray(cp.comp_sizes, dtype=np.uint64)
        out = np.empty(int(cp.raw_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=np.uint8)
        out_offs = np.ascontiguousarray(cp.raw_offs, dtype=np.uint64)
        out_caps = np.ascontiguousarray(cp.raw_sizes, dtype=np.uint64)
        nb = C.c_uint64(0)

        def b2(nthreads, passes):
            t = L.zref_time_oneshot_mt(cp.comp.ctypes.data, offs.ctypes.data, sizes.ctypes.data, nfiles, out.ctypes.data,
                                       out_offs.ctypes.data, out_caps.ctypes.data, nthreads, passes, C.byref(nb))
            assert nb.value == total_out * passes, "libzstd baseline failed"
            return t
        b2(cores, 1)
        end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
        ok = bool((out[:end] == cp.raw[:end]).all())
        # thread count: the fastest of {all hardware threads, half, a quarter} on a short probe (SMT rarely helps libzstd)
        probe = {}
        for nt in sorted({cores, max(cores // 2, 1), max(cores // 4, 1)}):
            # (freshly created threads take a few hundred ms to spread over the cores: each probe lasts >= ~1 s)
            p = max(1, -(-(256 << 20) // total_out))
            while True:
                tp = b2(nt, p)
                if tp >= 1.0 or p >= 1 << 16:
                    break
                p = int(p * max(2.0, 1.2 / max(tp, 1e-3)))
            probe[nt] = total_out * p / tp
        best_nt = max(probe, key=probe.get)
        reps = max(1, min(100000, int(budget_s * 0.6 * probe[best_nt] / total_out)))
        tt = b2(best_nt, reps)
        b2_gibs = total_out * reps / tt / GIB
        cores_used = best_nt
        scratch = np.empty(int(cp.raw_sizes.max()) + 64, dtype=np.uint8)

        def b1():
            return L.zref_time_stream8k(cp.comp.ctypes.data, offs.ctypes.data, sizes.ctypes.data, nfiles, scratch.ctypes.data,
                                        len(scratch), C.byref(nb))
        t1 = b1()
        reps1 = max(1, min(200, int(budget_s * 0.4 / max(t1, 1e-4))))
        tt1 = sum(b1() for _ in range(reps1))
        b1_gibs = total_out * reps1 / tt1 / GIB
        return {"value": round(b2_gibs, 3), "unit": "GiB/s", "cores": cores_used, "host_hw_threads": cores, "kind": "reference",
                "impl": "libzstd %s (system .so via dlopen = the reference's codec dependency; the Rust reference itself is unbuildable here)" % oracle.LibZstd.version(),
                "sample": "all %d files of the workload x %d passes, one-shot ZSTD_decompressDCtx, one file per task on %d threads created once (%.1f s; probe GiB/s by thread count: %s)" % (nfiles, reps, cores_used, tt, {k: round(v / GIB, 1) for k, v in probe.items()}),
                "stream8k_1thread": {"value": round(b1_gibs, 3), "unit": "GiB/s", "cores": 1,
                                     "sample": "same files x %d passes, ZSTD_decompressStream into an 8 KiB buffer (copy_decode shape) (%.1f s)" % (reps1, tt1)},
                "verified_equal": ok}
    # no libzstd on the box: time the oracle (the C restatement), one thread
    n = min(nfiles, 64)
    t0 = time.time(); done = 0
    while time.time() - t0 < budget_s:
        for i in range(n):
            rc, out = oracle.decode(cp.comp_file(i).tobytes(), cap=int(cp.raw_sizes[i]))
            assert rc == 0
            done += len(out)
    return {"value": round(done / (time.time() - t0) / GIB, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
            "sample": "first %d files, oracle/zstd_oracle.c through ctypes, repeated for %.0f s" % (n, budget_s)}


def cpu_baseline_dict(cp, dictionary, budget_s=10.0):
    """Config 5: libzstd with the shared dictionary digested ONCE (ZSTD_createDDict, outside the timed region) and
    ZSTD_decompress_usingDDict per record -- one record per task on a thread pool, a reused DCtx per thread."""
    import ctypes as C
    import oracle
    if not oracle.LibZstd.available():
        return {"value": None, "unit": "GiB/s", "cores": 1, "kind": "reference", "sample": "no libzstd on this host"}
    L = oracle.lib()
    cores = os.cpu_count() or 1
    nfiles = cp.nfiles
    total_out = int(cp.raw_sizes.sum())
    offs = np.ascontiguousarray(cp.comp_offs, dtype=np.uint64)
    sizes = np.ascontiguousarray(cp.comp_sizes, dtype=np.uint64)
    end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
    out = np.zeros(end + 64, dtype=np.uint8)  # (records are 16-byte aligned in the corpus image: the gaps are zero on both sides)
    out_offs = np.ascontiguousarray(cp.raw_offs, dtype=np.uint64)
    out_caps = np.ascontiguousarray(cp.raw_sizes, dtype=np.uint64)
    nb = C.c_uint64(0)
    dbytes = bytes(dictionary)

    def run(nthreads, passes):
        t = L.zref_time_oneshot_mt_dict(cp.comp.ctypes.data, offs.ctypes.data, sizes.ctypes.data, nfiles, out.ctypes.data,
                                        out_offs.ctypes.data, out_caps.ctypes.data, nthreads, passes, C.byref(nb), dbytes, len(dbytes))
        assert nb.value == total_out * passes, "libzstd dictionary baseline failed"
        return t
    run(cores, 1)
    ok = bool((out[:end] == cp.raw[:end]).all())
    # Each thread count is measured on a run of its own that lasts >= ~1 s (passes doubled until it does): freshly
    # created threads take a few hundred ms to spread over the cores, so short probes under-report by several times.
    per = max(budget_s / 6.0, 0.5)
    best = None; probe = {}
    for nt in sorted({cores, max(cores // 2, 1), max(cores // 4, 1)}, reverse=True):
        passes = max(1, -(-(64 << 20) // total_out))
        while True:
            tt = run(nt, passes)
            if tt >= per or passes >= 1 << 20:
                break
            passes = int(passes * max(2.0, 1.2 * per / max(tt, 1e-3)))
        rate = total_out * passes / tt
        probe[nt] = rate
        if best is None or rate > best[0]:
            best = (rate, nt, passes, tt)
    rate, best_nt, reps, tt = best
    return {"value": round(rate / GIB, 3), "unit": "GiB/s", "cores": best_nt, "host_hw_threads": cores, "kind": "reference",
            "impl": "libzstd %s ZSTD_decompress_usingDDict (dictionary digested once), system .so via dlopen" % oracle.LibZstd.version(),
            "sample": "all %d records of the workload x %d passes, one record per task on %d threads created once (%.1f s; GiB/s by thread count: %s)"
                      % (nfiles, reps, best_nt, tt, {k: round(v / GIB, 2) for k, v in probe.items()}),
            "verified_equal": ok}


def recorded_traffic(workload):
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(p)).get(workload)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--files", type=int, default=0, help="files per GPU (default 1000; 10000 for cfg4)")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import corpus
    import fuse_zstd_amd as mzd
    mzd.build()
    mzd.init([local_rank])  # raises when the HIP library / GPU is missing: no fallback

    kind, cfg_id, kind_mod, desc = WORKLOADS[args.workload]
    nfiles = args.files or {"cfg4": 10000, "cfg4lu": 10000, "cfg5": 50000}.get(args.workload, 1000)
    sizes = file_sizes(args.workload, nfiles, rank, world)
    dictionary, dict_id = None, 0
    if args.workload == "cfg5":  # trained on the first 4 000 records, 110 KiB cap; the same dictionary on every rank
        tr = np.random.RandomState(55).randint(300, 3001, size=4000)
        dictionary = corpus.train_dict(kind, cfg_id, [int(x) for x in tr], cap=112640)
        dict_id = mzd.load_dict(dictionary)
    cp = corpus.build_corpus(kind, cfg_id, sizes, first_index=rank, stride=world, level=args.level, kind_mod=kind_mod, dictionary=dictionary)
    C = int(cp.comp_sizes.sum()); U = int(cp.raw_sizes.sum())

    comp_d = torch.from_numpy(cp.comp).to(dev)  # cp.comp carries >= 64 bytes of zero padding (MZD_SRC_PADDING)
    out_offs = cp.raw_offs
    out_d = torch.zeros(int(out_offs[-1] + cp.raw_sizes[-1]) + 64, dtype=torch.uint8, device=dev)
    jobs = mzd.api.make_jobs([comp_d.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes,
                             [out_d.data_ptr() + int(o) for o in out_offs], cp.raw_sizes, [dict_id] * cp.nfiles if dict_id else None)
    batch = mzd.Batch(0, jobs)
    stream = torch.cuda.Stream(dev)  # a real (non-NULL) stream: the kernel and the timing events share it
    sp = stream.cuda_stream
    torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, 1)):
        batch.launch(sp)
    res = batch.collect(sp)
    bad = [(i, st, n) for i, (st, n) in enumerate(res) if st != 0 or n != int(cp.raw_sizes[i])]
    if bad:
        raise SystemExit("decode failed: %r" % bad[:5])
    got = out_d.cpu().numpy()
    end = int(out_offs[-1] + cp.raw_sizes[-1])
    verified = bool((got[:end] == cp.raw[:end]).all())  # raw gaps are zero on both sides
    if not verified:
        raise SystemExit("GPU output differs from the corpus bytes")

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b in evs:  # torch creates the HIP event on first record: do that outside the timed region
        a.record(stream); b.record(stream)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        evs[k][0].record(stream)
        batch.launch(sp)
        evs[k][1].record(stream)
    t_sub = time.perf_counter()
    fence()
    t1 = time.perf_counter()
    batch.collect(sp)
    elapsed = t1 - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / max(args.steps, 1)
    span_ms = evs[0][0].elapsed_time(evs[-1][1])  # first launch start -> last launch end, on the GPU's clock
    if os.environ.get("MZD_BENCH_GAPS"):
        gaps = [evs[k][1].elapsed_time(evs[k + 1][0]) for k in range(args.steps - 1)]
        sys.stderr.write("wall %.3f ms (submit loop %.3f ms), gpu span %.3f ms, gaps(ms) %s\n" % (elapsed * 1e3, (t_sub - t0) * 1e3, span_ms, " ".join("%.3f" % g for g in gaps)))
    last_ms = mzd.last_kernel_ms(0)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(U), float(C), kernel_ms], dtype=torch.float64, device=dev)
        allv = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(allv, tot)
        U_all = sum(float(v[0]) for v in allv); C_all = sum(float(v[1]) for v in allv)
    else:
        U_all, C_all = float(U), float(C)

    if rank == 0:
        value = U_all * args.steps / elapsed / GIB
        achieved = (C + U) / (kernel_ms * 1e-3) / 1e9  # this rank's kernel, GB/s
        line = {
            "metric": "decompressed GiB/s through the zstd decode path (device-resident; byte-exact vs libzstd)",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc), "files_per_gpu": nfiles, "files_total": nfiles * world,
                       "decompressed_bytes_per_gpu": U, "compressed_bytes_per_gpu": C, "zstd_level": args.level,
                       "sharding": "file i -> GPU i mod N, no collective", "compressor": "libzstd " + corpus.zstd_version()},
            "verified_byte_exact": verified,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_of_measured_copy_6290": round(achieved / 6290.0, 5),
                         "traffic": recorded_traffic(args.workload),
                         "kernel": "mzd_decode_kernel_tasks" if int(cp.raw_sizes.max()) > 131072 else "mzd_decode_kernel_files", "kernel_ms_avg": round(kernel_ms, 4), "kernel_ms_last_lib_events": round(last_ms, 4),
                         "algorithmic_bytes_per_launch": C + U},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cp, args.cpu_seconds) if dictionary is None else cpu_baseline_dict(cp, dictionary, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    batch.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
