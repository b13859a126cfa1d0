"""Summarise the two --pmc passes of tools/rocprof.sh (FETCH_SIZE, WRITE_SIZE; separate passes, as MI355X_MICROARCH.md prescribes)
into HBM bytes per bench step: python tools/pmc_summary.py gpurun_out/prof_<tag> <tag>  ->  gpurun_out/prof_<tag>_pmc.json
A bench step may be more than one kernel (the small-file kernel, then a general driver for what it hands on; the second
kernel's counters also carry the write-back of lines the first left dirty in L2): the step's traffic is the sum over the
step's kernels."""
import collections, csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
def counter(dirname, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            if r.get("Counter_Name") == name and ("mzd_decode_kernel" in kn or "mzd_small_kernel" in kn or "mzd_lds_kernel" in kn):
                acc[kn.split("(")[0].replace("void ", "").replace("mzd::", "")].append(float(r["Counter_Value"]))
    return acc
fe, wr = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
bench = json.load(open(out + "/bench.json"))
steps_f = max(len(v) for v in fe.values()); steps_w = max(len(v) for v in wr.values())
fetch_kib = sum(sum(v) for v in fe.values()) / steps_f
write_kib = sum(sum(v) for v in wr.values()) / steps_w
res = {"tag": tag, "steps_seen": [steps_f, steps_w],
       "per_kernel_KiB_per_step": {"FETCH_SIZE": {k: sum(v) / len(v) for k, v in fe.items()}, "WRITE_SIZE": {k: sum(v) / len(v) for k, v in wr.items()}},
       "FETCH_SIZE_KiB_per_step_raw": fetch_kib, "WRITE_SIZE_KiB_per_step_raw": write_kib,
       "algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_launch"]}
# MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
# coalesced streaming reads.  These kernels read with 4..16-byte accesses, which the guide calls uncalibrated: both the raw
# and the doubled figure are recorded.
res["hbm_bytes_per_step_raw"] = (fetch_kib + write_kib) * 1024
res["hbm_bytes_per_step_fetch_doubled"] = (2 * fetch_kib + write_kib) * 1024
res["traffic_over_algorithmic_raw"] = res["hbm_bytes_per_step_raw"] / res["algorithmic_bytes_per_step"]
json.dump(res, open(out + "/../prof_" + tag + "_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
