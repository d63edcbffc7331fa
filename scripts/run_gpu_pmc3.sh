# latency / instruction-mix counters of every k_cape_cin* kernel a python script launches: run_gpu_pmc3.sh <tag> <script.py> [args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_3.txt <<'X'
pmc: SQ_WAVES SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES
pmc: SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64
pmc: SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
X
rm -rf gpurun_out/pmc_$tag
timeout -k 10 500 rocprofv3 -i /tmp/pmc_3.txt --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/${tag}_pmc_run.log 2>&1; echo pmc rc=$?
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
