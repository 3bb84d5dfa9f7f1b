# bench.py --timed-only on the tile layouts of the --LD kernels, in turn, two rounds: the panel's own tiles, the site list's
# rows back to back (compact_align 1), every window on a tile boundary (compact_align 32), on a 4-row boundary:
#   bash tools/sweep_layouts.sh [bench flags]
for r in 1 2; do
  for o in "--opt compact_tiles=-1" "--opt compact_tiles=1 --opt compact_align=1" "--opt compact_tiles=1 --opt compact_align=32" "--opt compact_tiles=1 --opt compact_align=4"; do
    echo -n "$o: "; python bench.py --timed-only $o "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4), 'layout', d['ld_layout'])"
  done
done
