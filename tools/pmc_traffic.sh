#!/bin/bash
# HBM traffic of the engine's kernels: two separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: bash tools/pmc_traffic.sh <tag>
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "gpurun_out/$1_$c" -- \
      "$PY" bench.py --timed-only --steps 3 --warmup 1 $BENCH_FLAGS > "gpurun_out/$1_$c.log" 2>&1
done
"$PY" - "$1" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{tag}_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ibdg::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    out[c] = {k: sum(v) / len(v) for k, v in agg.items()}
print(json.dumps(out, indent=1))
open(f"gpurun_out/{tag}_traffic.json", "w").write(json.dumps(out, indent=1))
PY
