# Diagnostic: one GPU call's worth of checks for a change to the small-file kernel (mzd_lds.hip): phase stamps, corpora against
# the generator, the parity tests that reach the kernel, the mutation corpus.  TAG=name bash tools/small_cycle.sh
# (run as:  gpurun --timeout 900 -- 'TAG=x bash tools/small_cycle.sh')
TAG=${TAG:-x}
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python tools/lds_stamps.py cfg4 > gpurun_out/${TAG}_stamps.txt 2>&1
python tools/lds_check.py quick > gpurun_out/${TAG}_check.txt 2>&1
python tools/fuzz_small.py 11 > gpurun_out/${TAG}_fuzz.txt 2>&1
python tools/fuzz_small.py 12 >> gpurun_out/${TAG}_fuzz.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "small or config4 or config5 or both_drivers or dictionary or single_byte or golden" > gpurun_out/${TAG}_tests.txt 2>&1
tail -3 gpurun_out/${TAG}_tests.txt; grep -h "seed\|MISMATCH" gpurun_out/${TAG}_fuzz.txt; grep -h "TOTAL\|cfg4" gpurun_out/${TAG}_check.txt; grep -v amdgpu.ids gpurun_out/${TAG}_stamps.txt
