cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_cur
timeout -k 10 500 rocprofv3 -i profiles/pmc_counters.txt --kernel-trace --output-format csv -d gpurun_out/pmc_cur -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc.log 2>&1; echo pmc rc=$?
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_cur/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print('PMC', k, len(v), sum(v)/len(v))
PY
