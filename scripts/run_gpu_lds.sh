# bench + LDS counters of the headline kernel (per-wavefront means)
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1]); print('BENCH', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_small.txt <<'X'
pmc: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc: SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
X
rm -rf gpurun_out/pmc_lds
timeout -k 10 300 rocprofv3 -i /tmp/pmc_small.txt --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_lds/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn,d in agg.items():
    print('KERNEL', kn)
    for k,v in sorted(d.items()): print('  PMC', k, round(sum(v)/len(v)/16384,1))
PY
