# Same-box A/B of library builds over the all-outputs legs of scripts/run_gpu_configs.py: run_gpu_ab_configs.sh "c3 c5" lib1.so lib2.so ...
cfgs=$1; shift
rm -f gpurun_out/ab_cfg.log
for rep in 1 2 3; do for L in "$@"; do
  XPARCEL_LIB=$PWD/xarray_parcel_amd/lib/$L timeout -k 10 300 python scripts/run_gpu_configs.py $cfgs 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        for k, v in json.loads(l).items(): print('$L', k, round(v['kernel_ms'], 3))
" | tee -a gpurun_out/ab_cfg.log
done; done
python3 - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open('gpurun_out/ab_cfg.log'):
    lib, rest = l.split(' ', 1); k, v = rest.rsplit(' ', 1); d[(k, lib)].append(float(v))
for (k, lib), v in sorted(d.items()): print('MEDIAN %-34s %-16s %.3f  %s' % (k, lib, statistics.median(v), v))
PY
