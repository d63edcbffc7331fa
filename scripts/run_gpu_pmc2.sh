# like run_gpu_pmc_any.sh with a second counter set (LDS / fetch / wait breakdown)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_2.txt <<'X'
pmc: SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc: SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD
pmc: SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
X
rm -rf gpurun_out/pmc_$tag
timeout -k 10 500 rocprofv3 -i /tmp/pmc_2.txt --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/${tag}_pmc_run.log 2>&1; echo pmc rc=$?
python3 - $tag <<'PY' > gpurun_out/${tag}_pmc.txt
import csv, glob, collections, sys
tag = sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{tag}/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn,d in sorted(agg.items()):
    w = sum(d['SQ_WAVES'])/len(d['SQ_WAVES'])
    if w < 1000: continue
    print('KERNEL', kn, 'waves', int(w), 'launches', len(d['SQ_WAVES']))
    print('   ', {k: (round(sum(v)/len(v)/w,1) if k.startswith('SQ_') else round(sum(v)/len(v))) for k,v in sorted(d.items()) if k != 'SQ_WAVES'})
PY
cat gpurun_out/${tag}_pmc.txt
rm -rf gpurun_out/pmc_$tag
