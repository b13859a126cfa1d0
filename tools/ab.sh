# Diagnostic: bench.py's T1 number of several builds side by side on ONE box (boxes differ by up to 10 %), each build twice.
#   WL="cfg2 cfg4lu" SOS="libmzd_base.so libmzd.so" bash tools/ab.sh        (appends to gpurun_out/ab.txt as it goes)
mkdir -p gpurun_out
for w in $WL; do for so in $SOS $SOS; do
MZD_AB_SO=$so timeout -k 10 120 python tools/ab.py --workload $w --no-cpu-baseline --no-others --no-t2 --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w $so', d['value'], d['ms_per_step'], d['roofline']['achieved'])" | tee -a gpurun_out/ab.txt; done; done
