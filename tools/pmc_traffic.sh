# HBM traffic of the LD kernel: two separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  usage: bash tools/pmc_traffic.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/$1_$c -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$1_$c.log 2>&1
done
python - <<PY
import csv,glob,collections,json
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    agg=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/$1_%s/*/*counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            if 'ibdg::' in r['Kernel_Name']:
                agg[r['Kernel_Name'].split('(')[0][:60]].append(float(r['Counter_Value']))
    out[c]={k: sum(v)/len(v) for k,v in agg.items()}
print(json.dumps(out, indent=1))
open('gpurun_out/$1_traffic.json','w').write(json.dumps(out, indent=1))
PY
