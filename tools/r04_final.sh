#!/bin/bash
# The bench line and the kernel trace of the same call once the counter profiles it replays are committed, and the
# steps of 1/2, 1/4, 1/8 of the chromosome:  bash tools/r04_final.sh <tag>
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
timeout -k 10 900 "$PY" bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- "$PY" bench.py --timed-only > gpurun_out/${tag}_stats.log 2>&1 &&
cp gpurun_out/${tag}_stats/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv && echo "stats done" &&
timeout -k 10 300 "$PY" bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-many > gpurun_out/${tag}_bench_driver_flags.json 2>/dev/null &&
for n in 4000000 2000000 1000000 500000; do
  timeout -k 10 120 "$PY" bench.py --timed-only --sites $n 2>/dev/null | "$PY" -c "import sys,json; d=json.loads(sys.stdin.readline()); print($n, d['ms_per_step'], d['ld_launch_ms'])"
done > gpurun_out/${tag}_shard_steps.txt
rc=$?
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo "final rc=$rc"
