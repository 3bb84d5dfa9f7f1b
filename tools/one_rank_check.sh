# step clock, the engine's device clock and the side kernel's share of a step (args: --opt name=value ...)
for n in 500000 4000000; do
  python bench.py --steps 20 --warmup 5 --sites $n --no-cpu-baseline --no-e2e --no-many "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('sites $n', [(round(p['step_device_ms'],4), round(p['ms_per_step'],4), round(p['ld_launch_ms'],4)) for p in d['per_rank']], {k: round(v, 4) for k, v in d['kernel_ms'].items()})"
done
