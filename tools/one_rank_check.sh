for f in 1 0; do for n in 500000 4000000; do
  python bench.py --steps 20 --warmup 5 --sites $n --no-cpu-baseline --no-e2e --no-many --opt finalize_in_next=$f 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline())
print('fin_next $f sites $n', [(round(p['step_device_ms'],4), round(p['ms_per_step'],4), round(p['ld_launch_ms'],4)) for p in d['per_rank']], d['kernel_ms'])"
done; done
