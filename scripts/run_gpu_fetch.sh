# L2 fabric-side read counters per k_cape_cin kernel for one config share: run_gpu_fetch.sh <tag> <c2|c4|c5> [modes...]
# (calibration of FETCH_SIZE on this kernel's own access widths: 4 B / lane for float grids, 8 B / lane for double)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
printf 'pmc: FETCH_SIZE\npmc: TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum\npmc: SQ_INSTS_VMEM_RD SQ_WAVES\n' > /tmp/fetch_counters.txt
rm -rf gpurun_out/pmc_$tag
timeout -k 10 600 rocprofv3 -i /tmp/fetch_counters.txt --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/run_gpu_modes.py "$@" > gpurun_out/${tag}_fetch.log 2>&1; echo pmc rc=$?
python3 - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{tag}/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_cape_cin' in r['Kernel_Name']:
            agg[r['Kernel_Name'].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
with open(f'gpurun_out/{tag}_fetch.txt', 'w') as out:
    for k, d in sorted(agg.items()):
        line = k + ' ' + ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(d.items()))
        print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/pmc_$tag
