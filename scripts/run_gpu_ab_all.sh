# Same-box A/B of library builds over c2 / c4 share / c5 share (family mode): run_gpu_ab_all.sh lib1.so lib2.so ...   (libs under xarray_parcel_amd/lib/)
# three alternating rounds per config; prints every kernel time and the medians at the end
rm -f gpurun_out/ab_all.log
for cfg in c2 c4 c5; do for rep in 1 2 3; do for L in "$@"; do
  r=$(XPARCEL_LIB=$PWD/xarray_parcel_amd/lib/$L timeout -k 10 200 python scripts/run_gpu_modes.py $cfg family 2>/dev/null | tail -1)
  echo "$L $r" | tee -a gpurun_out/ab_all.log
done; done; done
python3 - <<'PY'
import json, collections, statistics
d = collections.defaultdict(list)
for l in open('gpurun_out/ab_all.log'):
    lib, js = l.split(' ', 1)
    for k, v in json.loads(js).items(): d[(k, lib)].append(v)
for (k, lib), v in sorted(d.items()): print('MEDIAN %-28s %-18s %.4f  %s' % (k, lib, statistics.median(v), v))
PY
