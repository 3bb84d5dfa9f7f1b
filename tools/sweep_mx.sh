for o in "$@"; do
  echo "== $o"; timeout -k 10 120 python bench.py --timed-only $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4), d['ld_layout'])"
done
