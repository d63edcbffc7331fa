# per-wavefront counters of the k_cape_cin kernel for one bench mode: run_gpu_pmc_mode.sh <bench args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_small.txt <<'X'
pmc: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc: SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pmc: FETCH_SIZE
pmc: WRITE_SIZE
X
rm -rf gpurun_out/pmc_mode
timeout -k 10 400 rocprofv3 -i /tmp/pmc_small.txt --kernel-trace --output-format csv -d gpurun_out/pmc_mode -- python3 bench.py --steps 2 --warmup 1 --no-cpu "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_mode/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob('gpurun_out/pmc_mode/pmc_1/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']: dur[r['Kernel_Name'][:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for kn,d in agg.items():
    print('KERNEL', kn, 'us', [round(x) for x in dur[kn]])
    w = sum(d['SQ_WAVES'])/len(d['SQ_WAVES'])
    for k,v in sorted(d.items()):
        m = sum(v)/len(v)
        print('  PMC', k, round(m/w,1) if k.startswith('SQ_') else round(m,1))
PY
