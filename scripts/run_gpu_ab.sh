# A/B of library variants in one box: bench (c2) + c5 share per variant; usage: run_gpu_ab.sh "" _old ...
for v in "$@"; do
  export XPARCEL_LIB=$GRAFT_REPO_ROOT/xarray_parcel_amd/lib/libxparcel$v.so
  for rep in 1 2; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('VARIANT', '[$v]', 'c2 kernel_ms', round(d['roofline']['kernel_ms'],4))"
  done
  timeout -k 10 600 python scripts/run_gpu_c5.py 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT', '[$v]', {k:(round(v['kernel_ms'],2), v['indices_match_sample']) for k,v in d.items()})"
done
