#!/bin/bash
# Usage: bash tools/pmc.sh "<counters>" [bench args]   -- one rocprofv3 --pmc pass, prints per-launch averages per mzd kernel
set -e
ctr="$1"; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_tmp; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out" -- python3 bench.py --no-cpu-baseline --no-others --no-t2 --no-traffic --steps 4 --warmup 2 "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "mzd_decode_kernel" in kn or "mzd_small_kernel" in kn or "mzd_lds_kernel" in kn:
            acc[(kn.split("(")[0].split("::")[-1][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, k), v in sorted(acc.items()):
    print("%-28s %-20s per launch %.4g  (launches %d)" % (kn, k, sum(v) / len(v), len(v)))
PY
