for o in "mx_counts=1" "mx_counts=0" "mx_counts=1 --opt compact_tiles=-1" "mx_counts=0 --opt compact_tiles=-1" "mx_counts=1 --opt windows_per_wave=12" "mx_counts=1 --opt windows_per_wave=8" "mx_counts=1 --opt ring_slots=3"; do
  echo "== $o"; timeout -k 10 120 python bench.py --timed-only --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4), d['ld_layout'])"
done
