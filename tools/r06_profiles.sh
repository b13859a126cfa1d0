#!/bin/bash
# Round-6 profile set (on the GPU box, from the repo root): bash tools/r06_profiles.sh [workloads...]
# Per workload: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (tools/rocprof.sh); then instruction counters (tools/pmc.sh).
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
ws="$@"; [ -z "$ws" ] && ws="cfg4 cfg2 cfg5 cfg3 cfg4lu big1m cfgmid cfg2x8 cfg4x4 cfg3x8"
for w in $ws; do
  extra=""
  case $w in cfg4lu|cfg2x8|cfg3x8|big1m|cfgmid) extra="--steps 6 --warmup 2";; esac
  bash tools/rocprof.sh r06_$w --workload $w $extra > gpurun_out/r06_prof_$w.log 2>&1 || echo "rocprof.sh $w failed"
  echo "done $w"
done
{
for w in cfg2 cfg4 cfg5 cfg3 cfg2x8 cfg4x4; do
  echo "== $w"
  bash tools/pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" --workload $w 2>&1 | grep "per launch"
done
echo "== cfg2x8: utilisation"
bash tools/pmc.sh "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" --workload cfg2x8 2>&1 | grep "per launch"
bash tools/pmc.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" --workload cfg2x8 2>&1 | grep "per launch"
bash tools/pmc.sh "SQC_ICACHE_REQ SQC_ICACHE_MISSES" --workload cfg2x8 2>&1 | grep "per launch"
} > gpurun_out/r06_pmc_instructions.txt
cat gpurun_out/r06_pmc_instructions.txt
