#!/bin/bash
# The chip's clock under a build's --LD kernel: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) over the kernel's
# duration in the same rocprofv3 pass, for every library named (IBDG_LIB):  bash tools/kernel_clock.sh "<lib> <lib> ..."
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
i=0
for lib in $1; do
  i=$((i + 1))
  export IBDG_LIB=$PWD/$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "gpurun_out/clk_p$i" -- \
      "$PY" bench.py --timed-only --steps 20 --warmup 5 > "gpurun_out/clk_p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "gpurun_out/clk_p$i.log"; }
  "$PY" - "gpurun_out/clk_p$i" "$lib" <<'PY'
import csv, glob, sys
d, lib = sys.argv[1], sys.argv[2]
cyc, dur = [], []
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_ld_popcount" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc.append(float(r["Counter_Value"]) / 8)
            if "Start_Timestamp" in r:
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
if not dur:
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_ld_popcount" in r["Kernel_Name"]:
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
c, t = sum(cyc) / len(cyc), sum(dur) / len(dur)
print(f"{lib}: {len(cyc)} launches, {c:.0f} cycles, {t / 1e3:.1f} us under the counter pass, {c / t:.3f} GHz")
PY
done
find gpurun_out -mindepth 1 -maxdepth 1 -type d -name "clk_p*" -exec rm -rf {} +
