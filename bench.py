#!/usr/bin/env python3
"""bench.py -- decompressed GiB/s of the zstd decode path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (the whole-file decode fuse-zstd runs on open(), reference
src/main.rs:463-467) over one batch of synthetic .zst files already resident in HBM: the timed
region contains only kernel launches (device-resident compressed bytes in, decompressed bytes
left in HBM).  Workload at N=1 = the corpus BASELINE.json's north_star names as the target ("decompressed GiB/s on a
10 000-file synthetic-JSON .zst corpus"; configs[3], the reference's own benchmarks/parallel-files.fio:3-7 shape): 10 000
independent 4 KiB JSON files written like the reference's writer (level 3, checksum, pledged size; src/main.rs:781-791);
configs[1] (1 000 x 128 KiB single-block frames: `cfg2`) is the first of `other_workloads`.  With N GPUs every rank takes
files r, r+N, r+2N, ... of an N x 10 000-file corpus (file i -> GPU i mod N, no collective; weak scaling).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4|cfg2|cfg3|cfg4lu|cfg5|big1m|cfgmid|...] [--files F]
                  [--no-others] [--no-t2] [--no-cpu-baseline]

ONE JSON line (rank 0).  `value` is T1 of SURVEY.md 8d: inputs in HBM when the timed region starts, outputs left in
HBM (several buffer sets are rotated so that the working set exceeds the 256 MiB Infinity Cache).  Extra objects:
  roofline         HBM bound.  achieved = algorithmic bytes per launch (compressed bytes read once + decompressed
                   bytes written once) / average kernel duration measured here with events on the
                   launch stream.  traffic = HBM bytes per launch, measured by this run where rocprofv3 is on the machine
                   (two child passes, FETCH_SIZE and WRITE_SIZE; --no-traffic skips them), else the value recorded in profiles/.
  cpu_baseline     rank 0 at N=1: the reference's CPU codec (system libzstd through dlopen) timed on this box's host
                   cores on the same files, repeated to ~10 s of CPU work.
  t2_end_to_end    T2: the path open() takes -- host buffers -> mzd_decode_batch -> host buffers, PCIe included
                   (the "through FUSE read path" half of BASELINE's metric; never `value`).
  other_workloads  the remaining BASELINE configurations (cfg2 1 000 x 128 KiB with its own T2, cfg3 Silesia-proxy, cfg4lu
                   log-uniform sizes, cfg5 shared dictionary, big1m 400 x 1 MiB multi-block files = the block-task driver),
                   each T1 with value / kernel_ms / roofline.frac / the kernels the library says it ran / byte-exact flag
                   (CPU baselines only for the small-file corpora: the default run must stay within minutes).
  single_file      one 1 MiB JSON file (BASELINE configs[0] shape): kernel ms, host-path ms, CPU streaming ms.
  per_rank         N > 1: every rank's own value and roofline.frac.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP starts: the host path overlaps 4 kernel streams and 2 copy streams (mzd_host.cpp)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured copy
GIB = float(1 << 30)
MALL_BYTES = 256 << 20

WORKLOADS = {
    # name: (corpus kind, cfg id, kind_mod, description)
    "cfg2": ("json", 2, 0, "1000 x 128 KiB single-block JSON frames, zstd level 3, checksum+FCS (BASELINE configs[1])"),
    "cfg3": ("text", 3, 7, "Silesia-proxy mix as 128 KiB single-block frames, level 3 (BASELINE configs[2]; Silesia itself is not on the box)"),
    "cfg4": ("json", 4, 0, "4 KiB JSON files, parallel-files.fio shape (BASELINE configs[3])"),
    "cfg4lu": ("json", 4, 0, "JSON files log-uniform 4 KiB..1 MiB (BASELINE configs[3] variant)"),
    "cfg5": ("json", 5, 0, "JSON records of 300..3000 B, one shared ZDICT-trained dictionary (BASELINE configs[4]; 50 000 files per GPU by default)"),
    # the same generators with enough files to fill the machine several times over: the sustained rate of the kernel, where the
    # headline configurations measure one launch's latency (1 000 files on 1 024 workgroup slots; 10 000 small files: two rounds of groups)
    "cfg2x8": ("json", 2, 0, "8 000 x 128 KiB single-block JSON frames (cfg2's generator, eight times the files: the sustained single-block rate)"),
    "cfg4x4": ("json", 4, 0, "40 000 x 4 KiB JSON files (cfg4's generator, four times the files: the sustained small-file rate)"),
    "cfg3x8": ("text", 3, 7, "8 000 x 128 KiB frames of cfg3's seven-class mix (eight times the files: chains of very different length, handed out longest first)"),
    # between the two kernels' sweet spots: too big for the small-file kernel (> 8 KiB), every file a single block or two
    "cfgmid": ("json", 4, 0, "10 000 JSON files log-uniform 8 KiB..128 KiB (above the small-file kernel's limit, at most one block: a workgroup per file, ten rounds)"),
    # few big files: every file's blocks on different workgroups (the block-task driver)
    "big1m": ("json", 1, 0, "400 x 1 MiB JSON files of eight blocks each, level 3, checksum (BASELINE configs[0]'s file, 400 of them: block tasks)"),
}
DEFAULT_FILES = {"cfg2": 1000, "cfg3": 1000, "cfg4": 10000, "cfg4lu": 10000, "cfg5": 50000, "cfg2x8": 8000, "cfg4x4": 40000, "cfg3x8": 8000, "big1m": 400, "cfgmid": 10000}


def file_sizes(workload, nfiles, rank, world):
    if workload in ("cfg2", "cfg3", "cfg2x8", "cfg3x8"):
        return [131072] * nfiles
    if workload in ("cfg4", "cfg4x4"):
        return [4096] * nfiles
    if workload == "big1m":
        return [1 << 20] * nfiles
    if workload == "cfgmid":
        rng = np.random.RandomState(4321)
        allsz = np.exp(rng.uniform(np.log(8193), np.log(131072), size=nfiles * world)).astype(np.int64)
        return [int(x) for x in allsz[rank::world]]
    if workload == "cfg5":
        rng = np.random.RandomState(55)
        allsz = rng.randint(300, 3001, size=nfiles * world)
        return [int(x) for x in allsz[rank::world]]
    rng = np.random.RandomState(1234)  # same table on every rank; rank r takes entries r, r+world, ...
    allsz = np.exp(rng.uniform(np.log(4096), np.log(1 << 20), size=nfiles * world)).astype(np.int64)
    return [int(x) for x in allsz[rank::world]]


def silesia_chunks(nfiles, rank, world):
    """Pieces of 128 KiB of the files under $SILESIA_DIR (sorted by name), piece i -> rank i mod world, at most nfiles per rank;
    None when the variable is not set or the directory holds nothing (then the generator's Silesia-proxy mix is used)."""
    d = os.environ.get("SILESIA_DIR")
    if not d or not os.path.isdir(d):
        return None
    out, i = [], 0
    for fn in sorted(os.listdir(d)):
        path = os.path.join(d, fn)
        if not os.path.isfile(path):
            continue
        with open(path, "rb") as f:
            while True:
                piece = f.read(131072)
                if not piece:
                    break
                if i % world == rank and len(out) < nfiles:
                    out.append(piece)
                i += 1
    return out or None


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
    """HBM bytes per step from the last committed counter passes (tools/rocprof.sh; not measured in this run: the PMC passes need
    rocprofv3 around the process).  profiles/pmc_traffic.json carries the commit and date they were taken at ("_recorded_at")."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(p)).get(workload)
    except (OSError, ValueError):
        return None


def measure_traffic(workload, files, level):
    """HBM bytes per launch of this workload, measured NOW: two child runs of this script under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes: they do not fit one pass on gfx950), the
    counters of the step's kernels summed per launch.  Raw counter values (these kernels read with 4..16-byte accesses, for which
    the guide calls FETCH_SIZE uncalibrated).  None when rocprofv3 is not on the machine; {"failed": why} when a pass fails (then the
    line carries the recorded value of profiles/pmc_traffic.json and says so).  The child is a plain one-GPU run of this script: the
    launcher's variables (RANK, WORLD_SIZE, MASTER_*, TORCHELASTIC_*) are taken out of its environment."""
    import csv, glob, shutil, subprocess, tempfile
    if not shutil.which("rocprofv3"):
        return None
    out = {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT") and not k.startswith("TORCHELASTIC_")}
    env["TMPDIR"] = "/tmp"
    tmp = tempfile.mkdtemp(prefix="mzd_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--workload", workload, "--files", str(files), "--level", str(level), "--steps", "4", "--warmup", "1",
                   "--no-others", "--no-t2", "--no-cpu-baseline", "--no-traffic"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
            if r.returncode != 0:
                return {"failed": "%s pass: exit %d: %s" % (ctr, r.returncode, (r.stderr or "")[-300:])}
            per_kernel = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    kn = row.get("Kernel_Name", "")
                    if row.get("Counter_Name") == ctr and ("mzd_decode_kernel" in kn or "mzd_lds_kernel" in kn):
                        per_kernel.setdefault(kn.split("(")[0], []).append(float(row["Counter_Value"]))
            if not per_kernel:
                return {"failed": "%s pass: no counter rows for the decode kernels" % ctr}
            launches = max(len(v) for v in per_kernel.values())
            out[ctr] = sum(sum(v) for v in per_kernel.values()) / launches * 1024.0  # KiB -> bytes, per launch of the batch
    except Exception as e:
        return {"failed": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fb, wb = int(out["FETCH_SIZE"]), int(out["WRITE_SIZE"])
    return {"bytes": fb + wb, "fetch_bytes": fb, "write_bytes": wb}


def traffic_recorded_at():
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(p)).get("_recorded_at")
    except (OSError, ValueError):
        return None


class Workload:
    """One corpus, device-resident, with enough rotating buffer sets to exceed the Infinity Cache."""

    def __init__(self, name, nfiles, rank, world, level, dev, mzd, corpus):
        import torch
        self.name, self.mzd = name, mzd
        kind, cfg_id, kind_mod, self.desc = WORKLOADS[name]
        self.nfiles = nfiles
        sizes = file_sizes(name, nfiles, rank, world)
        self.dictionary, self.dict_id = None, 0
        if name == "cfg5":  # trained on the first 4 000 records, 110 KiB cap; the same dictionary on every rank
            tr = np.random.RandomState(55).randint(300, 3001, size=4000)
            self.dictionary = corpus.train_dict(kind, cfg_id, [int(x) for x in tr], cap=112640)
            self.dict_id = mzd.load_dict(self.dictionary)
        chunks = silesia_chunks(nfiles, rank, world) if name == "cfg3" else None
        if chunks:  # the real Silesia files when the box has them (SILESIA_DIR), each 128 KiB piece its own frame
            self.cp = cp = corpus.build_corpus_from_chunks(chunks, level=level)
            self.nfiles = len(chunks)
            self.real_data = True
            self.desc = "Silesia (SILESIA_DIR) as 128 KiB single-block frames, level %d (BASELINE configs[2])" % level
        else:
            self.cp = cp = corpus.build_corpus(kind, cfg_id, sizes, first_index=rank, stride=world, level=level, kind_mod=kind_mod, dictionary=self.dictionary)
        self.C = int(cp.comp_sizes.sum()); self.U = int(cp.raw_sizes.sum())
        self.end = int(cp.raw_offs[-1] + cp.raw_sizes[-1])
        self.nsets = max(1, min(8, -(-(MALL_BYTES + (64 << 20)) // (self.C + self.U))))
        self.sets = []
        comp_h = torch.from_numpy(cp.comp)  # cp.comp carries >= 64 bytes of zero padding (MZD_SRC_PADDING)
        for _ in range(self.nsets):
            comp_d = comp_h.to(dev)
            out_d = torch.zeros(self.end + 64, dtype=torch.uint8, device=dev)
            jobs = mzd.api.make_jobs([comp_d.data_ptr() + int(o) for o in cp.comp_offs], cp.comp_sizes,
                                     [out_d.data_ptr() + int(o) for o in cp.raw_offs], cp.raw_sizes,
                                     [self.dict_id] * cp.nfiles if self.dict_id else None)
            self.sets.append((comp_d, out_d, mzd.Batch(0, jobs)))

    def check(self, sp, what):
        """Every job of every set: status, length, bytes (raw gaps are zero on both sides)."""
        cp = self.cp
        for comp_d, out_d, batch in self.sets:
            res = batch.collect(sp)
            bad = [(i, st, n) for i, (st, n) in enumerate(res) if st != 0 or n != int(cp.raw_sizes[i])]
            if bad:
                raise SystemExit("%s: decode failed %s: %r" % (self.name, what, bad[:5]))
            got = out_d.cpu().numpy()
            if not bool((got[:self.end] == cp.raw[:self.end]).all()):
                raise SystemExit("%s: GPU output differs from the corpus bytes %s" % (self.name, what))
            # A launch of small files alone ends with the small-file kernel; what that kernel hands on is decoded when the batch is
            # COLLECTED, outside the timed launches -- counter word 4 counts those files, and a timed rate must not leave work out
            self.handed_on = max(getattr(self, "handed_on", 0), int(self.mzd.debug_counters(0)[4]))
        return True

    def zero_outputs(self):
        for _, out_d, _ in self.sets:
            out_d.zero_()

    def free(self):
        for _, _, batch in self.sets:
            batch.free()
        self.sets = []


def time_t1(w, steps, warmup, stream, fence):
    """K launches, set k % nsets each; returns (wall seconds, average kernel ms from events on the launch stream)."""
    import torch
    sp = stream.cuda_stream
    for k in range(max(warmup, 1) * w.nsets):
        w.sets[k % w.nsets][2].launch(sp)
    w.check(sp, "before timing")
    w.zero_outputs()  # what is verified after the timed region was written inside it
    # The kernels' time: ONE pair of HIP events on the launch stream around the K launches (average = span / K, the gaps between
    # two launches included).  Round 4 recorded a pair per step, and the library a pair of its own per launch: four packets between two
    # kernels, ~10 us of a 0.3 ms step that measured the measurement.  The launches are untimed now (MZD_LAUNCH_UNTIMED).
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); e1.record(stream)  # (torch creates the HIP event on first record: outside the timed region)
    fence()
    t0 = time.perf_counter()
    e0.record(stream)
    for k in range(steps):
        w.sets[k % w.nsets][2].launch(sp, untimed=True)
    e1.record(stream)
    fence()
    t1 = time.perf_counter()
    kernel_ms = e0.elapsed_time(e1) / max(steps, 1)
    w.kernels = w.mzd.last_kernel_name(0)  # what the library launched (mzd_last_kernel_name), dominant kernel first
    if steps >= w.nsets:
        w.check(sp, "after the timed region")  # every status again, and the bytes the timed launches wrote
    if "mzd_decode_kernel" not in w.kernels and getattr(w, "handed_on", 0) * 1000 > w.cp.nfiles:  # (a handful is noise; the count is in the line: "handed_on")
        raise SystemExit("%s: the small-file kernel handed %d files on to the general driver, which decodes them at collect -- outside the timed region" % (w.name, w.handed_on))
    return t1 - t0, kernel_ms


def t2_end_to_end(w, mzd, reps=5):
    """Host buffers -> mzd_decode_batch -> host buffers (PCIe both ways, staging included), as open() would call it.
    Three shapes: caller's buffers from mzd_host_alloc (pinned: no staging copy), ordinary pageable buffers, and two
    calls in flight from two threads (what a multi-threaded daemon sustains)."""
    cp = w.cp
    L = mzd.api.lib()
    n = cp.nfiles
    dids = [w.dict_id] * n if w.dict_id else None
    out = {}

    def run(jobs, check=True):
        t0 = time.perf_counter()
        rc = L.mzd_decode_batch(jobs, n)
        dt = time.perf_counter() - t0
        assert rc == 0, "T2 decode failed"
        if check:  # (outside the timed region; the threaded leg checks once at its end: this loop holds the interpreter lock)
            assert all(j.status == 0 for j in jobs), "T2 decode failed"
        return dt

    def bench(src_arr, dst_arr, label):
        jobs = mzd.api.make_jobs([src_arr.ctypes.data + int(o) for o in cp.comp_offs], cp.comp_sizes,
                                 [dst_arr.ctypes.data + int(o) for o in cp.raw_offs], cp.raw_sizes, dids)
        run(jobs)
        dst_arr[:w.end] = 0
        ts = [run(jobs) for _ in range(reps)]
        ok = bool((dst_arr[:w.end] == cp.raw[:w.end]).all())
        best = min(ts)
        out[label] = {"ms": round(best * 1e3, 3), "value": round(w.U / best / GIB, 2), "unit": "GiB/s", "ms_all": [round(t * 1e3, 2) for t in ts],
                      "kernel_ms_sum_over_chunks": round(mzd.last_kernel_ms(0), 3), "byte_exact": ok}
        return jobs

    pin_in = mzd.HostBuffer(len(cp.comp)); pin_in.a[:] = cp.comp
    pin_out = mzd.HostBuffer(w.end + 64)
    pin_out2 = mzd.HostBuffer(w.end + 64)
    try:
        jobs_a = bench(pin_in.a, pin_out.a, "pinned")
        page_out = np.zeros(w.end + 64, dtype=np.uint8)
        bench(cp.comp, page_out, "pageable")
        # two calls in flight (two threads, the batch twice): steady-state rate of a caller that keeps the device fed
        jobs_b = mzd.api.make_jobs([pin_in.a.ctypes.data + int(o) for o in cp.comp_offs], cp.comp_sizes,
                                   [pin_out2.a.ctypes.data + int(o) for o in cp.raw_offs], cp.raw_sizes, dids)
        # (round 6: 40 rounds a thread behind one untimed call of each job set -- the second set's first call allocates its staging buffers,
        #  and about once in 25-30 calls a thread BOTH threads lose 7-8 ms at the same moment with the GPU idle (profiles/r06_t2_pairs.txt: a
        #  stall on the host side of the runtime, seen with two submitting threads only): four rounds measured 18 or 33 GiB/s by luck.  The
        #  figure is the average over everything, stalls included; the per-call median and the number of stalled calls stand beside it.)
        run(jobs_b)
        rounds = 40
        per_call = [[], []]

        def worker(k, jobs):
            for _ in range(rounds):
                per_call[k].append(run(jobs, check=False))
        ths = [threading.Thread(target=worker, args=(k, j)) for k, j in enumerate((jobs_a, jobs_b))]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        ok = all(j.status == 0 for j in jobs_a) and all(j.status == 0 for j in jobs_b) and bool((pin_out2.a[:w.end] == cp.raw[:w.end]).all())
        calls = sorted(per_call[0] + per_call[1])
        med = calls[len(calls) // 2]
        out["pinned_two_calls_in_flight"] = {"ms_per_batch": round(dt / (2 * rounds) * 1e3, 3), "value": round(2 * rounds * w.U / dt / GIB, 2), "unit": "GiB/s", "byte_exact": ok,
                                             "rounds_per_thread": rounds, "call_ms_median": round(med * 1e3, 3), "calls_over_twice_the_median": sum(1 for t in calls if t > 2 * med),
                                             "call_ms_max": round(calls[-1] * 1e3, 2)}
    finally:
        pin_in.free(); pin_out.free(); pin_out2.free()
    out["what"] = ("T2 (SURVEY.md 8d): host buffers -> mzd_decode_batch -> host buffers on %s, one call = one batch; "
                   "best of %d calls; the link is PCIe Gen5 x16" % (w.name, reps))
    return out


def single_file(mzd, corpus, dev, stream):
    """One 1 MiB JSON file written like the reference's writer (BASELINE configs[0] shape: 8 blocks, 7 treeless):
    what ONE open() costs.  kernel: device-resident; host: mzd_decode on host buffers; cpu: libzstd streaming, 1 thread."""
    import torch
    cp = corpus.build_corpus("json", 1, [1 << 20])
    comp_d = torch.from_numpy(cp.comp).to(dev)
    out_d = torch.zeros((1 << 20) + 64, dtype=torch.uint8, device=dev)
    jobs = mzd.api.make_jobs([comp_d.data_ptr()], cp.comp_sizes, [out_d.data_ptr()], cp.raw_sizes)
    b = mzd.Batch(0, jobs)
    sp = stream.cuda_stream
    ks = []
    for _ in range(6):
        b.launch(sp)
        res = b.collect(sp)
        ks.append(mzd.last_kernel_ms(0))
    ok = res[0] == (0, 1 << 20) and bool((out_d.cpu().numpy()[:1 << 20] == cp.raw[:1 << 20]).all())
    b.free()
    src = cp.comp_file(0).tobytes()
    hs = []
    for _ in range(6):
        t0 = time.perf_counter()
        st, got = mzd.decode(src, 1 << 20)
        hs.append((time.perf_counter() - t0) * 1e3)
    ok = ok and st == 0 and got == cp.raw_file(0).tobytes()
    r = {"file": "1 MiB JSON, level 3, checksum (8 blocks)", "kernel_ms": round(min(ks[1:]), 3), "host_path_ms": round(min(hs[1:]), 3), "byte_exact": ok}
    try:
        import oracle
        if oracle.LibZstd.available():
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                got = oracle.LibZstd.decompress(src, 1 << 20, stream8k=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            r["cpu_stream8k_ms"] = round(min(ts), 3)
    except Exception:  # the CPU leg is informative only
        pass
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--files", type=int, default=0, help="files per GPU (default: 10000 for cfg4/cfg4lu; 1000 for cfg2/cfg3; 50000 for cfg5; 400 for big1m)")
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the other_workloads / single_file objects")
    ap.add_argument("--no-t2", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # Rehearsal of the N > 1 flow on a box with ONE GPU (tools/, never the driver): every rank on device 0, collectives over gloo
    # on CPU tensors.  The value of such a run means nothing (the ranks share the card); what it checks is the protocol.
    rehearsal = world > 1 and os.environ.get("MZD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev  # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import corpus
    import fuse_zstd_amd as mzd
    mzd.build()
    mzd.init([local_rank])  # raises when the HIP library / GPU is missing: no fallback
    if os.environ.get("MZD_DRIVER"):  # diagnostic (tools/): force one kernel choice, see mzd_debug_set_driver
        mzd.set_driver(int(os.environ["MZD_DRIVER"]))

    stream = torch.cuda.Stream(dev)  # a real (non-NULL) stream: the kernels and the timing events share it

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def local_fence():
        torch.cuda.synchronize(dev)

    nfiles = args.files or DEFAULT_FILES[args.workload]
    w = Workload(args.workload, nfiles, rank, world, args.level, dev, mzd, corpus)
    torch.cuda.synchronize(dev)
    elapsed, kernel_ms = time_t1(w, args.steps, args.warmup, stream, fence)
    my_value = w.U * args.steps / elapsed / GIB
    my_frac = (w.C + w.U) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(w.U), float(w.C), kernel_ms, my_value, my_frac], dtype=torch.float64, device=cdev)
        allv = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(allv, tot)
        U_all = sum(float(v[0]) for v in allv)
        per_rank = [{"rank": r, "value": round(float(v[3]), 3), "kernel_ms": round(float(v[2]), 4), "roofline_frac": round(float(v[4]), 5)} for r, v in enumerate(allv)]
    else:
        U_all, per_rank = float(w.U), None

    line = None
    if rank == 0:
        value = U_all * args.steps / elapsed / GIB
        achieved = (w.C + w.U) / (kernel_ms * 1e-3) / 1e9  # this rank's kernel, GB/s
        line = {
            "metric": "decompressed GiB/s through FUSE read path; byte-exact vs libzstd",
            "value": round(value, 3), "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic" if not getattr(w, "real_data", False) else "files of SILESIA_DIR",
            "value_is": "T1: the decode of open() (reference src/main.rs:463-467) with compressed files resident in HBM when the timed region starts and decoded files left in HBM; the host-to-host rate of the same path is t2_end_to_end",
            "config": {"workload": "%s: %s" % (args.workload, w.desc), "files_per_gpu": w.nfiles, "files_total": w.nfiles * world,
                       "decompressed_bytes_per_gpu": w.U, "compressed_bytes_per_gpu": w.C, "zstd_level": args.level,
                       "sharding": "file i -> GPU i mod N, no collective", "compressor": "libzstd " + corpus.zstd_version(),
                       "buffer_sets_rotated": w.nsets, "working_set_bytes": w.nsets * (w.C + w.U)},
            "verified_byte_exact": True,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_of_measured_copy_6290": round(achieved / 6290.0, 5),
                         "traffic": recorded_traffic(args.workload), "traffic_recorded_at": traffic_recorded_at(),
                         "kernel": w.kernels, "kernel_ms_avg": round(kernel_ms, 4), "handed_on": int(getattr(w, "handed_on", 0)),
                         "algorithmic_bytes_per_launch": w.C + w.U},
        }
        if per_rank:
            line["per_rank"] = per_rank

    # ---- the rest of the measurement contract: rank 0 at N = 1 only (the driver's scaling runs stay short)
    if world == 1:
        if not args.no_t2:
            line["t2_end_to_end"] = t2_end_to_end(w, mzd)
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w.cp, args.cpu_seconds) if w.dictionary is None else cpu_baseline_dict(w.cp, w.dictionary, args.cpu_seconds)
        w.free()
        del w
        torch.cuda.empty_cache()
        if not args.no_others:
            others = {}
            for name in ("cfg2", "cfg4", "cfg3", "cfg4lu", "cfg5", "big1m", "cfgmid", "cfg2x8", "cfg4x4", "cfg3x8"):
                if name == args.workload:
                    continue
                ow = Workload(name, DEFAULT_FILES[name], 0, 1, args.level, dev, mzd, corpus)
                steps = 10 if name not in ("cfg4lu", "cfg2x8", "cfg3x8", "big1m", "cfgmid") else 6
                el, kms = time_t1(ow, steps, 2, stream, local_fence)
                ach = (ow.C + ow.U) / (kms * 1e-3) / 1e9
                others[name] = {"workload": ow.desc, "files": ow.nfiles, "value": round(ow.U * steps / el / GIB, 3), "unit": "GiB/s",
                                "files_per_s": round(ow.nfiles * steps / el), "ms_per_step": round(el / steps * 1e3, 4), "kernel_ms": round(kms, 4),
                                "kernel": ow.kernels, "roofline_frac": round(ach / HBM_PEAK_GBS, 5), "achieved_GBps": round(ach, 2),
                                "algorithmic_bytes_per_launch": ow.C + ow.U, "traffic": recorded_traffic(name), "buffer_sets_rotated": ow.nsets, "byte_exact": True, "steps": steps}
                if name in ("cfg4", "cfg2") and not args.no_t2:
                    others[name]["t2_end_to_end"] = t2_end_to_end(ow, mzd, reps=3)
                if name in ("cfg2", "cfg4", "cfg5") and not args.no_cpu_baseline:  # the CPU beside them (short samples): the many-small-files corpora, and configs[1], whose host -> host rate INTEGRATION.md compares with it
                    others[name]["cpu_baseline"] = cpu_baseline(ow.cp, 2.0) if ow.dictionary is None else cpu_baseline_dict(ow.cp, ow.dictionary, 3.0)
                ow.free()
                del ow
                torch.cuda.empty_cache()
            line["other_workloads"] = others
            line["single_file"] = single_file(mzd, corpus, dev, stream)
            line["single_file_ms"] = line["single_file"]["kernel_ms"]
        if not args.no_traffic:  # roofline.traffic measured by this run (two rocprofv3 --pmc child passes over the headline workload)
            torch.cuda.empty_cache()
            m = measure_traffic(args.workload, nfiles, args.level)
            line["roofline"]["traffic_measured"] = bool(m) and "failed" not in m
            if m and "failed" in m:
                line["roofline"]["traffic_measure_failed"] = m["failed"]  # (the recorded value of profiles/pmc_traffic.json stands in)
            elif m:
                r = line["roofline"]
                r["traffic"] = m["bytes"]
                r["traffic_fetch_bytes"], r["traffic_write_bytes"] = m["fetch_bytes"], m["write_bytes"]
                r["traffic_over_algorithmic"] = round(m["bytes"] / r["algorithmic_bytes_per_launch"], 3)
                r["traffic_recorded_at"] = "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over child runs of this script, raw counter bytes per launch"
            # ... and the two workloads with the worst ratios (the block pipeline's queues: cfg2; the block tasks' byte maps: big1m), so that those
            # are this run's numbers too and not the record of profiles/pmc_traffic.json (two more pairs of child passes, ~20 s)
            if not args.no_others and line["roofline"].get("traffic_measured"):
                for name in ("cfg2", "big1m"):
                    ow = line.get("other_workloads", {}).get(name)
                    if ow is None:
                        continue
                    m2 = measure_traffic(name, DEFAULT_FILES[name], args.level)
                    ow["traffic_measured"] = bool(m2) and "failed" not in m2
                    if m2 and "failed" not in m2:
                        ow["traffic"] = m2["bytes"]
                        ow["traffic_fetch_bytes"], ow["traffic_write_bytes"] = m2["fetch_bytes"], m2["write_bytes"]
                        ow["traffic_over_algorithmic"] = round(m2["bytes"] / ow["algorithmic_bytes_per_launch"], 3)
                    elif m2:
                        ow["traffic_measure_failed"] = m2["failed"]
    else:
        w.free()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
