# timing sweep of the exponent-counting LD kernel.  IBDG_DEBUG (1 = skip window math, 2 = skip counting) is only
# honoured by an ablation build:  make -C ibdgem_amd/csrc clean all CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -DIBDG_TIMING_EXPERIMENT=1"
for ns in 4 8; do for g in 8 16 32; do for d in 0 3; do
IBDG_DEBUG=$d timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --opt ring_slots=$ns --opt windows_per_wave=$g --opt record_lds_bytes=90000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ns=$ns g=$g debug=$d', d['kernel_ms']['ld'])"
done; done; done
