for v in "$@"; do
  export XPARCEL_LIB=$GRAFT_REPO_ROOT/xarray_parcel_amd/lib/libxparcel$v.so
  for rep in 1 2; do
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --moist family 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('VARIANT', '[$v]', 'family c2 kernel_ms', round(d['roofline']['kernel_ms'],4))"
  done
done
