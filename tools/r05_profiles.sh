#!/bin/bash
# Round-5 profile set (on the GPU box, from the repo root): bash tools/r05_profiles.sh
# Per workload: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (tools/rocprof.sh), then instruction counters (tools/pmc.sh).
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
for w in cfg4 cfg2 cfg5 cfg3 cfg4lu big1m cfgmid cfg2x8 cfg4x4 cfg3x8; do
  extra=""
  [ "$w" = cfg4lu ] && extra="--steps 6 --warmup 2"
  [ "$w" = cfg2x8 ] && extra="--steps 6 --warmup 2"
  [ "$w" = cfg3x8 ] && extra="--steps 6 --warmup 2"
  [ "$w" = big1m ] && extra="--steps 6 --warmup 2"
  [ "$w" = cfgmid ] && extra="--steps 6 --warmup 2"
  bash tools/rocprof.sh r05_$w --workload $w $extra > gpurun_out/r05_prof_$w.log 2>&1 || echo "rocprof.sh $w failed"
  echo "done $w"
done
{
for w in cfg2 cfg4 cfg5 cfg3 cfg2x8 cfg4x4; do
  echo "== $w"
  bash tools/pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" --workload $w 2>&1 | grep "per launch"
done
} > gpurun_out/r05_pmc_instructions.txt
echo "== cfg4: LDS pipeline" >> gpurun_out/r05_pmc_instructions.txt
bash tools/pmc.sh "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" --workload cfg4 2>&1 | grep "per launch" >> gpurun_out/r05_pmc_instructions.txt
bash tools/pmc.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" --workload cfg4 2>&1 | grep "per launch" >> gpurun_out/r05_pmc_instructions.txt
cat gpurun_out/r05_pmc_instructions.txt
