#!/bin/bash
# per-kernel times of the steps of 1/8 of the chromosome (bench.py --timed-only --sites 500000) under rocprofv3 --stats
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shard_stats -- "$PY" bench.py --timed-only --sites 500000 "$@" > gpurun_out/shard_stats.log 2>&1
cp gpurun_out/shard_stats/*/*kernel_stats.csv gpurun_out/shard_kernel_stats.csv
cp gpurun_out/shard_stats/*/*kernel_trace.csv gpurun_out/shard_kernel_trace.csv 2>/dev/null
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
tail -1 gpurun_out/shard_stats.log
