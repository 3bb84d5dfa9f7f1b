for f in 1 0 1 0; do
  BENCH_FORCE_DEVICE=0 BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --sites 500000 --no-cpu-baseline --no-e2e --many-targets 20 --opt finalize_in_next=$f 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('fin_next $f', [(round(p['step_device_ms'],4), round(p['ms_per_step'],4), round(p['ld_launch_ms'],4)) for p in d['per_rank']])"
done
