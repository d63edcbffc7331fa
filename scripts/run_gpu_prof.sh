# Profile the bench command at the current sources: rocprofv3 --kernel-trace --stats summary + PMC passes (separate runs),
# written under gpurun_out/ as <tag>_stats.txt and <tag>_pmc.json ready to be copied into profiles/.
# usage: run_gpu_prof.sh <tag> [bench args...]      e.g.  run_gpu_prof.sh r02_a_family --moist family
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o cur -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-table-leg --no-config-legs "$@" > gpurun_out/${tag}_bench.log 2>&1; echo prof rc=$?
db=$(ls gpurun_out/prof_$tag/*/*.db gpurun_out/prof_$tag/*.db 2>/dev/null | head -1)
python3 profiles/summarize.py "$db" gpurun_out/${tag}_stats.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-table-leg --no-config-legs $*" > /dev/null
head -5 gpurun_out/${tag}_stats.txt
timeout -k 10 400 rocprofv3 -i profiles/pmc_counters.txt --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --preroll-seconds 0 --no-cpu --no-table-leg --no-config-legs "$@" > gpurun_out/${tag}_pmc.log 2>&1; echo pmc rc=$?
python3 - "$tag" "$@" <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
from xarray_parcel_amd import _lib
tag = sys.argv[1]; args = sys.argv[2:]
bench = json.loads([l for l in open(f'gpurun_out/{tag}_pmc.log') if l.startswith('{')][-1])
kernel = bench['roofline']['kernel']
agg = collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/pmc_{tag}/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].replace('void ', '').startswith(kernel):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
mean = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
out = {'command': 'rocprofv3 -i profiles/pmc_counters.txt --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --preroll-seconds 0 --no-cpu --no-table-leg --no-config-legs ' + ' '.join(args),
       'kernel': kernel, 'shape': [bench['config']['levels'], bench['config']['columns_this_rank']], 'csrc_sha': _lib.csrc_sha(),
       'per_launch_mean': mean,
       'hbm_read_bytes_corrected': mean.get('FETCH_SIZE', 0) * 1024 * 2, 'hbm_write_bytes': mean.get('WRITE_SIZE', 0) * 1024,
       'algorithmic_bytes': bench['roofline']['algorithmic_bytes_per_launch'],
       'note': 'FETCH_SIZE x 1024 x 2 (gfx950 half-count correction per MI355X_MICROARCH.md, calibrated in round 1: reproduces the bytes of '
               'a streaming read); WRITE_SIZE x 1024'}
out['hbm_traffic_bytes'] = out['hbm_read_bytes_corrected'] + out['hbm_write_bytes']
json.dump(out, open(f'gpurun_out/{tag}_pmc.json', 'w'), indent=1)
w = mean.get('SQ_WAVES', 1)
print(kernel, 'traffic/algorithmic', out['hbm_traffic_bytes'] / out['algorithmic_bytes'])
for k, v in mean.items():
    print(' ', k, round(v / w, 1) if k.startswith('SQ_') and k != 'SQ_WAVES' else round(v, 1))
PY
