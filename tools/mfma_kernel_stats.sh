#!/bin/bash
# per-kernel times of one run over T comparison individuals: bash tools/mfma_kernel_stats.sh <tag> [T] [IBDG_OPTS]
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
tag=${1:-r04}; T=${2:-60}; export IBDG_OPTS=${3:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_mfma_stats -- "$PY" tools/multi_target.py 4000000 $T > gpurun_out/${tag}_mfma_stats.log 2>&1
cp gpurun_out/${tag}_mfma_stats/*/*kernel_stats.csv gpurun_out/${tag}_mfma_kernel_stats.csv
rm -rf gpurun_out/${tag}_mfma_stats
"$PY" - "$tag" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(f"gpurun_out/{sys.argv[1]}_mfma_kernel_stats.csv")):
    m = re.search(r"ibdg::(?:\(anonymous namespace\)::)?(k_\w+)", r["Name"])
    if m and int(r["Calls"]) <= 64:
        print("%-24s calls %4s avg %9.1f us  total %9.1f us" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
grep T= gpurun_out/${tag}_mfma_stats.log
