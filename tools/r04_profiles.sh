#!/bin/bash
# All counter / trace passes of round 4 in one call on the GPU box:  bash tools/r04_profiles.sh <tag>
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
# 1. the bench line, and in the same call on the same box the per-kernel times of the timed steps
timeout -k 10 900 "$PY" bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- "$PY" bench.py --timed-only > gpurun_out/${tag}_stats.log 2>&1 &&
cp gpurun_out/${tag}_stats/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv && echo "stats done" &&
timeout -k 10 300 "$PY" bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-many > gpurun_out/${tag}_bench_driver_flags.json 2>/dev/null &&
# 2. counters of the dominant kernel (six passes) and its traffic (two)
bash tools/pmc_ld.sh ${tag}pmc "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
    "SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
    "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" "LdsUtil MeanOccupancyPerCU SALUBusy VALUBusy" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" > gpurun_out/${tag}pmc_summary.txt 2>&1 && echo "pmc done" &&
bash tools/pmc_traffic.sh ${tag} > gpurun_out/${tag}_traffic_summary.txt 2>&1 && echo "traffic done" &&
# 3. the matrix-core kernel: times over T (the panel's own tiles and the compacted ones), kernel stats and counters at T = 15
"$PY" tools/multi_target.py 4000000 1 2 3 4 5 8 15 16 30 60 120 500 > gpurun_out/${tag}_multi_target.txt 2>&1 &&
IBDG_OPTS=compact_tiles=1 "$PY" tools/multi_target.py 4000000 1 15 60 500 > gpurun_out/${tag}_multi_target_compacted.txt 2>&1 &&
bash tools/mfma_kernel_stats.sh ${tag} 60 > gpurun_out/${tag}_mfma_kernel_stats.txt 2>&1 &&
bash tools/pmc_any.sh ${tag}mfma "tools/multi_target.py 4000000 15" "LdsUtil MfmaUtil SALUBusy VALUBusy GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES MeanOccupancyPerCU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES" > gpurun_out/${tag}mfma_summary.txt 2>&1 && echo "mfma done" &&
# 4. the site preparation, per kernel
bash tools/prep_kernel_stats.sh ${tag} > gpurun_out/${tag}_prep_summary.txt 2>&1 &&
# 5. the micro-benchmarks of the round: the 64-bit shift hazard, the counts on the matrix cores
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/shift64 shift64_top_vgpr.hip 2>/dev/null && timeout -k 5 60 /tmp/shift64 > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_shift64_top_vgpr.txt") &&
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fp4_count fp4_count.hip 2>/dev/null && timeout -k 5 60 /tmp/fp4_count > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_fp4_count.txt") &&
# 6. one comparison individual: the vector-ALU counts beside the matrix-core ones, the panel's own tiles beside the compacted ones
bash tools/sweep_mx.sh "--opt mx_counts=1" "--opt mx_counts=0" "--opt mx_counts=1 --opt compact_tiles=-1" "--opt mx_counts=0 --opt compact_tiles=-1" "--opt mx_counts=1 --opt sum_dpp=0" "--opt mx_counts=1" "--opt mx_counts=0" > gpurun_out/${tag}_count_units.txt 2>&1
rc=$?
# only the summaries travel back (the raw traces are tens of megabytes per pass)
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo "profiles rc=$rc"
