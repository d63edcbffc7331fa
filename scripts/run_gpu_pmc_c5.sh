# per-wavefront counters of the three c5 kernels (MU / ML / SB, fp32 input, 100 levels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_small.txt <<'X'
pmc: SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pmc: SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR
pmc: FETCH_SIZE
X
rm -rf gpurun_out/pmc_c5
timeout -k 10 500 rocprofv3 -i /tmp/pmc_small.txt --kernel-trace --output-format csv -d gpurun_out/pmc_c5 -- python3 scripts/run_gpu_c5.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_c5/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn,d in sorted(agg.items()):
    w = sum(d['SQ_WAVES'])/len(d['SQ_WAVES'])
    print('KERNEL', kn, 'waves', int(w))
    print('   ', {k: (round(sum(v)/len(v)/w,1) if k.startswith('SQ_') else round(sum(v)/len(v))) for k,v in sorted(d.items()) if k != 'SQ_WAVES'})
PY
