#!/bin/bash
# PMC passes over the --LD kernels of bench.py's workload (run on the GPU box):
#     bash tools/pmc_ld.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]
# One rocprofv3 run per pass (--kernel-trace --pmc only: this pool refuses --pmc together with the
# runtime/sys trace domains).  Per-kernel averages of every counter go to gpurun_out/<tag>_pmc.json.
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is unset)}"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
# the interpreter itself after `--`: a PATH shim or an env wrapper would be an exec after the profiler's
# preloaded library has initialised the GPU, which this pool forbids
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
i=0
for counters in "$@"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "gpurun_out/${tag}_p$i" -- \
      "$PY" bench.py --timed-only --steps 3 --warmup 1 $BENCH_FLAGS > "gpurun_out/${tag}_p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "gpurun_out/${tag}_p$i.log"; }
done
"$PY" - "$tag" <<'PY'
import collections, csv, glob, json, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{tag}_p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ibdg::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches_averaged": max(len(v) for v in cs.values())}
       for k, cs in agg.items()}
json.dump(out, open(f"gpurun_out/{tag}_pmc.json", "w"), indent=1)
for k, cs in out.items():
    if "ld_popcount" in k:
        print(k, json.dumps(cs))
PY
