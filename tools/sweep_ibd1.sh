#!/bin/bash
# The timed steps of bench.py over the launch geometry of the counting kernel (one box, one call):
#   bash tools/sweep_ibd1.sh            -> ring slots x windows per wave x LDS budget of a run's records
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() { printf "%-60s " "$*"; python bench.py --timed-only --steps 20 --warmup 5 "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4f ms/step  launch %.4f" % (d["ms_per_step"], d["ld_launch_ms"]))'; }
run
run --opt ibd0_after=0
for rs in 2 3 4; do run --opt ring_slots=$rs; done
for rs in 3 4; do for w in 8 12 16 24; do run --opt ring_slots=$rs --opt windows_per_wave=$w; done; done
run --opt ring_slots=3 --opt record_lds_bytes=8192
run --opt ring_slots=3 --opt record_lds_bytes=6144
run
