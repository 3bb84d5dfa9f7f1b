# timing sweep of the exponent-counting LD kernel over ring depth and windows per run
# (run on the GPU box: bash tools/sweep_ring.sh [rows])
rows=${1:-4000000}
for ns in 2 3 4 8; do for g in 8 16 24; do
timeout -k 10 300 python bench.py --sites $rows --no-cpu-baseline --opt ring_slots=$ns --opt windows_per_wave=$g --opt record_lds_bytes=40000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ring=$ns windows_per_run=$g', round(d['ms_per_step'],4), round(d['kernel_ms']['ld'],4))"
done; done
