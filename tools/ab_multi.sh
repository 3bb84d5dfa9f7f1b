# A/B of two builds of the library on one box: tools/multi_target.py with IBDG_LIB alternating
for r in 1 2 3; do
  for lib in build/libibdgem_hip_prev.so ibdgem_amd/libibdgem_hip.so; do
    echo "== $lib"; IBDG_LIB=$PWD/$lib python tools/multi_target.py 4000000 15 60 2>&1 | grep -v amdgpu
  done
done
