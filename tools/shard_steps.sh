# the timed step (bench.py --timed-only: ms per step, --LD launch ms) on the whole chromosome and on 1/2, 1/4, 1/8 of it: what a
# rank of a 1/2/4/8-GPU run does.   bash tools/shard_steps.sh [bench flags]
for n in 4000000 2000000 1000000 500000; do
  python bench.py --timed-only --sites $n "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print($n, round(d['ms_per_step'],4), round(d['ld_launch_ms'],4), 'layout', d['ld_layout'])"
done
