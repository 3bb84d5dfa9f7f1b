#!/bin/bash
# bench.py --timed-only at other window lengths (one box, one call): the panel's own tiles against the default (compacted after
# 22 runs, IBD1 form after 8), and the default with the form that counts everything.    bash tools/other_windows.sh
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
cd "$GRAFT_REPO_ROOT" || exit 1
echo "# bench.py --timed-only --window W: ms per step, --LD launch ms, layout, G sites/s"
for W in 32 50 100 200; do
  for o in "--opt compact_tiles=-1" "" "--opt ibd0_after=0"; do
    printf "window %-4s %-26s " "$W" "$o"
    python bench.py --timed-only --steps 20 --warmup 5 --window $W $o 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4f %.4f layout %d %.3f G sites/s" % (d["ms_per_step"], d["ld_launch_ms"], d["ld_layout"], d["value"]/1e9))'
  done
done
