for n in 500000 4000000; do
 for o in "guided_runs=4" "guided_runs=1" "guided_runs=2" "guided_runs=8" "guided_runs=16" "guided_runs=4 --opt windows_per_wave=12" "guided_runs=4 --opt windows_per_wave=8" "guided_runs=8 --opt windows_per_wave=12" "guided_runs=4 --opt windows_per_wave=24" "guided_runs=4"; do
  echo -n "sites $n $o: "; timeout -k 10 120 python bench.py --timed-only --sites $n --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['ld_launch_ms'],4))"
 done
done
