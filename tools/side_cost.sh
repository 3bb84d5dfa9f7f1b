#!/bin/bash
# bench.py --timed-only with and without a new individual per step / the per-row results: what the side kernels cost a step
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() { printf "%-50s " "$*"; python bench.py --timed-only --steps 40 --warmup 5 "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4f ms/step  launch %.4f" % (d["ms_per_step"], d["ld_launch_ms"]))'; }
for i in 1 2; do
run
run --same-target
run --opt site_results=0
run --same-target --opt site_results=0
done
