#!/bin/bash
# All evidence of round 5 in one call on the GPU box, at one commit:  bash tools/r05_profiles.sh <tag>
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
# 1. the driver's command: the line it parses and the detail file; then the per-kernel times of the SAME flags' timed steps
timeout -k 10 900 "$PY" bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err && cp gpurun_out/bench_detail.json gpurun_out/${tag}_bench_detail.json && echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- "$PY" bench.py --timed-only --steps 20 --warmup 5 > gpurun_out/${tag}_stats.log 2>&1 &&
cp gpurun_out/${tag}_stats/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv && tail -1 gpurun_out/${tag}_stats.log > gpurun_out/${tag}_timed_only_under_rocprof.json && echo "stats done" &&
timeout -k 10 300 "$PY" bench.py --timed-only --steps 20 --warmup 5 > gpurun_out/${tag}_timed_only.json 2>/dev/null &&
# 2. counters of the dominant kernel on the layout of the timed steps (compacted from the upload on, so that every dispatch the
#    passes average over is of that layout) and its traffic (two passes)
BENCH_FLAGS="--opt compact_tiles=1" bash tools/pmc_ld.sh ${tag}pmc "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
    "SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
    "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM" "LdsUtil MeanOccupancyPerCU SALUBusy VALUBusy" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" > gpurun_out/${tag}pmc_summary.txt 2>&1 && echo "pmc done" &&
BENCH_FLAGS="--opt compact_tiles=1" bash tools/pmc_traffic.sh ${tag} > gpurun_out/${tag}_traffic_summary.txt 2>&1 && echo "traffic done" &&
# 3. the matrix-core kernel: times over T on the panel's own tiles and on the compacted ones, kernel stats and counters at T = 15
"$PY" tools/multi_target.py 4000000 1 2 3 4 5 8 12 15 16 17 19 30 31 32 60 120 500 > gpurun_out/${tag}_multi_target.txt 2>&1 &&
IBDG_OPTS=compact_tiles=1 "$PY" tools/multi_target.py 4000000 1 4 15 16 30 60 500 > gpurun_out/${tag}_multi_target_compacted.txt 2>&1 &&
bash tools/mfma_kernel_stats.sh ${tag} 60 > gpurun_out/${tag}_mfma_kernel_stats.txt 2>&1 &&
bash tools/pmc_any.sh ${tag}mfma "tools/multi_target.py 4000000 15" "LdsUtil MfmaUtil SALUBusy VALUBusy GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES MeanOccupancyPerCU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/${tag}mfma_summary.txt 2>&1 && echo "mfma done" &&
# 4. the steps of 1/2, 1/4, 1/8 of the chromosome, the tile layouts, other windows
bash tools/shard_steps.sh > gpurun_out/${tag}_shard_steps.txt 2>&1 &&
bash tools/sweep_layouts.sh > gpurun_out/${tag}_layouts.txt 2>&1
rc=$?
find gpurun_out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo "profiles rc=$rc"
