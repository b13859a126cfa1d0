#!/bin/bash
# Usage: bash tools/pmc_kind.sh "<counters>" kind n   -- one rocprofv3 --pmc pass over tools/kind_one.py, per-launch averages per mzd kernel
set -e
ctr="$1"; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_tmp; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out" -- python3 tools/kind_one.py "$@" > "$out/run.log" 2> "$out/run.err" || { tail -5 "$out/run.err"; exit 1; }
cat "$out/run.log"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "mzd_decode_kernel" in kn or "mzd_lds_kernel" in kn:
            acc[(kn.split("(")[0].split("::")[-1][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, k), v in sorted(acc.items()):
    print("%-28s %-24s per launch %.4g  (launches %d)" % (kn, k, sum(v) / len(v), len(v)))
PY
