#!/bin/bash
# Usage: bash tools/pairs_pmc.sh "<counters>" [probe args]  -- one rocprofv3 --pmc pass over tools/pairs_probe.py; per-launch averages by workgroup size
set -e
ctr="$1"; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_pairs; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out" -- python3 tools/pairs_probe.py "$@" > "$out/probe.txt" 2> "$out/probe.err" || { tail -5 "$out/probe.err"; exit 1; }
cat "$out/probe.txt"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r.get("Kernel_Name", "")
        if "mzd_decode_kernel" in kn:
            wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")); grid = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            acc[(wg, grid, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (wg, grid, k), v in sorted(acc.items()):
    print("wg %-5s grid %-8s %-22s per launch %.4g  (launches %d)" % (wg, grid, k, sum(v) / len(v), len(v)))
PY
