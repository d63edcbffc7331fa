cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_small.txt <<'X'
pmc: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc: SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM
X
rm -rf gpurun_out/pmc_fam
timeout -k 10 300 rocprofv3 -i /tmp/pmc_small.txt --kernel-trace --output-format csv -d gpurun_out/pmc_fam -- python3 bench.py --steps 2 --warmup 1 --no-cpu --moist ${1:-family} > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_fam/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn,d in agg.items():
    print('KERNEL', kn)
    for k,v in sorted(d.items()): print('  PMC', k, round(sum(v)/len(v)/16384,1))
for f in glob.glob('gpurun_out/pmc_fam/pmc_1/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']: print('  DUR', r['Kernel_Name'][:50], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us')
PY
