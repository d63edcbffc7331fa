cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_small.txt <<'X'
pmc: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
X
for n in 64 128; do
rm -rf gpurun_out/pmc_n$n
timeout -k 10 300 rocprofv3 -i /tmp/pmc_small.txt --kernel-trace --output-format csv -d gpurun_out/pmc_n$n -- python3 bench.py --steps 2 --warmup 1 --no-cpu --nlev $n > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_n$n/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print('PMC$n', k, sum(v)/len(v)/16384)
PY
done
