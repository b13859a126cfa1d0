#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools_rocprof.sh <tag> [bench args...]
# Writes the rocprofv3 kernel-trace stats of `python3 bench.py ...` under gpurun_out/prof_<tag>/ and a
# short text summary gpurun_out/prof_<tag>.txt (copy that into profiles/ to commit it).
set -e
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline $*"
  echo "# bench line:"; cat "$out/bench.json"
  echo "# kernel stats (Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev):"
  head -12 "$f"
} > "$PWD/gpurun_out/prof_$tag.txt"
cat "$PWD/gpurun_out/prof_$tag.txt"
