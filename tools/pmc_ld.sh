# PMC pass over the LD kernel: usage tools/pmc_ld.sh <outdir> 0 "<counters>"   (second argument unused, kept for old command lines)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $3 --output-format csv -d gpurun_out/$1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$1.log 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/$1/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_ld_popcount' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('$1', {k: sum(v)/len(v) for k,v in agg.items()})
PY
