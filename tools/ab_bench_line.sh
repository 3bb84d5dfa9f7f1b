# A/B of two builds of the library on one box with the driver's own command (no --timed-only): value, ms per step, the
# dominant kernel's time and roofline.frac of every line.  bash tools/ab_bench_line.sh [rounds]
for r in $(seq 1 ${1:-3}); do
  for lib in build/libibdgem_hip_prev.so ibdgem_amd/libibdgem_hip.so; do
    echo -n "$lib: "; IBDG_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-many 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4g' % d['value'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))"
  done
done
