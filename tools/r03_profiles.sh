#!/bin/bash
# All counter / trace passes of a round in one call on the GPU box:  bash tools/r03_profiles.sh <tag>
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
# 1. per-kernel times of the timed steps
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- "$PY" bench.py --timed-only > gpurun_out/${tag}_stats.log 2>&1 &&
cp gpurun_out/${tag}_stats/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv &&
# 2. counters of the dominant kernel (four passes) and its traffic (two)
bash tools/pmc_ld.sh ${tag}pmc "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
    "SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
    "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" "LdsUtil MeanOccupancyPerCU SALUBusy VALUBusy" > gpurun_out/${tag}pmc_summary.txt 2>&1 &&
bash tools/pmc_traffic.sh ${tag} > gpurun_out/${tag}_traffic_summary.txt 2>&1 &&
# 3. the matrix-core kernel: times over T, kernel stats and counters at T = 15
"$PY" tools/multi_target.py 4000000 1 4 5 8 15 16 30 60 120 500 > gpurun_out/${tag}_multi_target.txt 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_mfma_stats -- "$PY" tools/multi_target.py 4000000 15 > gpurun_out/${tag}_mfma_stats.log 2>&1 &&
cp gpurun_out/${tag}_mfma_stats/*/*kernel_stats.csv gpurun_out/${tag}_mfma_kernel_stats.csv &&
bash tools/pmc_any.sh ${tag}mfma "tools/multi_target.py 4000000 15" "LdsUtil MfmaUtil SALUBusy VALUBusy GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES MeanOccupancyPerCU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" > gpurun_out/${tag}mfma_summary.txt 2>&1 &&
# 4. the site preparation, per kernel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prep_stats -- "$PY" tools/prep_times.py > gpurun_out/${tag}_prep_times.txt 2>&1 &&
cp gpurun_out/${tag}_prep_stats/*/*kernel_stats.csv gpurun_out/${tag}_prep_kernel_stats.csv
rc=$?
# only the summaries travel back (the raw traces are tens of megabytes per pass)
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo "profiles rc=$rc"
