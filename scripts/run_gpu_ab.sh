for v in "" _w3 _w4; do XPARCEL_LIB=$GRAFT_REPO_ROOT/xarray_parcel_amd/lib/libxparcel$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('VARIANT', '$v', 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['check'])"; done
