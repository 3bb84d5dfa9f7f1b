# A/B of two builds of the library on one box: bench.py --timed-only with IBDG_LIB alternating (args: extra bench flags)
for r in 1 2 3; do
  for lib in build/libibdgem_hip_prev.so ibdgem_amd/libibdgem_hip.so; do
    echo -n "$lib: "; IBDG_LIB=$PWD/$lib python bench.py --timed-only "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4))"
  done
done
