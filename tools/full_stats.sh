#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/full_stats.sh  ->  gpurun_out/full_kernel_stats.txt
# rocprofv3 --kernel-trace --stats over the plain `python3 bench.py` (the command the driver runs).
export TMPDIR=/tmp
rm -rf gpurun_out/full_kt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/full_kt -- python3 bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err || { tail -5 gpurun_out/full_bench.err; exit 1; }
f=$(find gpurun_out/full_kt -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py   (the command the driver runs: headline cfg2, then T2, CPU baseline, cfg3 / cfg4 / cfg4lu / cfg5, one big file)"
  echo "# mzd_decode_kernel_files is shared by cfg2 (0.88 ms per launch), cfg3 (1.36 ms), the host-path chunks and the empty launches behind the"
  echo "# small-file kernel: its average here is a mix; the headline workload alone is profiles/r02_cfg2_kernel_stats.txt"
  echo "# bench line:"; cat gpurun_out/full_bench.json
  echo "# kernel stats (Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev):"
  head -12 "$f"
} > gpurun_out/full_kernel_stats.txt
tail -12 gpurun_out/full_kernel_stats.txt | cut -c1-150
