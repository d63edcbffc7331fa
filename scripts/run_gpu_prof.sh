cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cur gpurun_out/pmc_cur
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cur -o cur -- python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/bench_prof.log 2>&1; echo prof rc=$?
grep '^{' gpurun_out/bench_prof.log | tail -1 > gpurun_out/bench_prof.json
timeout -k 10 300 rocprofv3 -i profiles/pmc_counters.txt --kernel-trace --output-format csv -d gpurun_out/pmc_cur -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc.log 2>&1; echo pmc rc=$?
python3 - <<'PY'
import csv, glob, collections, json
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_cur/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
out={k: sum(v)/len(v) for k,v in sorted(agg.items())}
json.dump(out, open('gpurun_out/pmc_cur.json','w'), indent=1)
print(out.get('FETCH_SIZE'), out.get('WRITE_SIZE'))
PY
timeout -k 10 200 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-300
