# A/B of several builds of the library on one box: tools/multi_target.py with IBDG_LIB in turn, three rounds
#   bash tools/ab_multi.sh "<lib> <lib> ..." [T ...]
libs="$1"; shift
for r in 1 2 3; do
  for lib in $libs; do
    echo "== $lib"; IBDG_LIB=$PWD/$lib python tools/multi_target.py 4000000 "${@:-60}" 2>&1 | grep -v amdgpu
  done
done
