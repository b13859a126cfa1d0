#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/rocprof.sh <tag> [bench args...]
# Three separate rocprofv3 passes over `python3 bench.py --no-cpu-baseline --no-others --no-t2 ...` (the headline workload alone,
# so that per-kernel averages are this workload's) (counters never share a
# pass with each other: FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950):
#   1. --kernel-trace --stats           -> gpurun_out/prof_<tag>.txt          (kernel durations)
#   2. --kernel-trace --pmc FETCH_SIZE  -> gpurun_out/prof_<tag>_pmc.json     (HBM traffic per launch)
#   3. --kernel-trace --pmc WRITE_SIZE
# Copy the .txt / .json into profiles/ to commit them.
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out/kt" "$out/fetch" "$out/write"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --no-cpu-baseline --no-others --no-t2 "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
f=$(find "$out/kt" -name '*kernel_stats.csv' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-others --no-t2 $*"
  echo "# bench line:"; cat "$out/bench.json"
  echo "# kernel stats (Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev):"
  head -8 "$f"
} > "$PWD/gpurun_out/prof_$tag.txt"
cat "$PWD/gpurun_out/prof_$tag.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --no-cpu-baseline --no-others --no-t2 "$@" > "$out/bench_fetch.json" 2> "$out/bench_fetch.err" || { tail -5 "$out/bench_fetch.err"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --no-cpu-baseline --no-others --no-t2 "$@" > "$out/bench_write.json" 2> "$out/bench_write.err" || { tail -5 "$out/bench_write.err"; exit 1; }
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
def counter(dirname, name):
    vals = []
    for f in glob.glob(out + "/" + dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("mzd_decode_kernel" in r.get("Kernel_Name", "") or "mzd_small_kernel" in r.get("Kernel_Name", "")) and r.get("Counter_Name") == name:
                vals.append(float(r["Counter_Value"]))
    return vals
fe, wr = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
bench = json.load(open(out + "/bench.json"))
res = {"tag": tag, "launches_seen": [len(fe), len(wr)],
       "FETCH_SIZE_KiB_per_launch_raw": sum(fe) / max(len(fe), 1), "WRITE_SIZE_KiB_per_launch_raw": sum(wr) / max(len(wr), 1),
       "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
# MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
# coalesced streaming reads.  This kernel reads with 4..16-byte accesses, which the guide calls uncalibrated:
# both the raw and the doubled figure are recorded.
res["hbm_bytes_per_launch_raw"] = (res["FETCH_SIZE_KiB_per_launch_raw"] + res["WRITE_SIZE_KiB_per_launch_raw"]) * 1024
res["hbm_bytes_per_launch_fetch_doubled"] = (2 * res["FETCH_SIZE_KiB_per_launch_raw"] + res["WRITE_SIZE_KiB_per_launch_raw"]) * 1024
json.dump(res, open(out + "/../prof_" + tag + "_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
