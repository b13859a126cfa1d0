#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/rocprof.sh <tag> [bench args...]
# Three separate rocprofv3 passes over `python3 bench.py --no-cpu-baseline --no-others --no-t2 --no-traffic ...` (the headline workload alone,
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
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --no-cpu-baseline --no-others --no-t2 --no-traffic "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
f=$(find "$out/kt" -name '*kernel_stats.csv' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-others --no-t2 --no-traffic $*"
  echo "# bench line:"; cat "$out/bench.json"
  echo "# kernel stats (Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev):"
  head -8 "$f"
} > "$PWD/gpurun_out/prof_$tag.txt"
cat "$PWD/gpurun_out/prof_$tag.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --no-cpu-baseline --no-others --no-t2 --no-traffic "$@" > "$out/bench_fetch.json" 2> "$out/bench_fetch.err" || { tail -5 "$out/bench_fetch.err"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --no-cpu-baseline --no-others --no-t2 --no-traffic "$@" > "$out/bench_write.json" 2> "$out/bench_write.err" || { tail -5 "$out/bench_write.err"; exit 1; }
python3 tools/pmc_summary.py "$out" "$tag"
