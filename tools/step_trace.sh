#!/bin/bash
# kernel trace (start / end of every dispatch) of bench.py --timed-only under rocprofv3: where a step's time goes between
# the kernels.   bash tools/step_trace.sh <tag> [bench flags]   -> gpurun_out/<tag>_kernel_trace.csv, <tag>_kernel_stats.csv, <tag>_gaps.txt
: "${GRAFT_REPO_ROOT:?run through gpurun}"
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- "$PY" bench.py --timed-only "$@" > gpurun_out/${tag}_trace.log 2>&1
cp gpurun_out/${tag}_trace/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
cp gpurun_out/${tag}_trace/*/*kernel_trace.csv gpurun_out/${tag}_kernel_trace.csv 2>/dev/null
rm -rf gpurun_out/${tag}_trace
tail -1 gpurun_out/${tag}_trace.log
python3 tools/step_gaps.py gpurun_out/${tag}_kernel_trace.csv > gpurun_out/${tag}_gaps.txt
cat gpurun_out/${tag}_gaps.txt
