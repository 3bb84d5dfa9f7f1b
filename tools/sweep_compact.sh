for o in "ring_slots=2" "ring_slots=3" "ring_slots=4" "ring_slots=8" "ring_slots=2 --opt windows_per_wave=8" "ring_slots=2 --opt windows_per_wave=32" "ring_slots=3 --opt windows_per_wave=32"; do
  echo "== $o"; timeout -k 10 120 python bench.py --timed-only --opt compact_tiles=1 --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4), d['ld_layout'])"
done
echo "== in place"; timeout -k 10 120 python bench.py --timed-only --opt compact_tiles=-1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4), d['ld_layout'])"
