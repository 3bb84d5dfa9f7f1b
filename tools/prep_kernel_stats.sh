#!/bin/bash
# per-kernel times of the site preparation (rocprofv3 --kernel-trace --stats over tools/prep_times.py): bash tools/prep_kernel_stats.sh <tag>
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prep_stats -- "$PY" tools/prep_times.py > gpurun_out/${tag}_prep_times.txt 2>&1
cp gpurun_out/${tag}_prep_stats/*/*kernel_stats.csv gpurun_out/${tag}_prep_kernel_stats.csv
rm -rf gpurun_out/${tag}_prep_stats
"$PY" - "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_prep_kernel_stats.csv")):
    if "k_prep" in r["Name"] or "gather" in r["Name"]:
        n = r["Name"].split("(")[0]
        print("%-34s calls %4s avg %8.1f us" % (n[-32:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 gpurun_out/${tag}_prep_times.txt
