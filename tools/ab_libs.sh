# bench.py --timed-only with several builds of the library in turn on one box (IBDG_LIB), two rounds:
#   bash tools/ab_libs.sh "<lib> <lib> ..." [bench flags]
libs="$1"; shift
for r in 1 2; do
  for lib in $libs; do
    echo -n "$lib: "; IBDG_LIB=$PWD/$lib python bench.py --timed-only "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4))"
  done
done
