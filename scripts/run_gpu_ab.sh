# Same-box A/B of library builds (boxes differ by ~3 %): run_gpu_ab.sh [bench args] -- lib1.so lib2.so ...
# (libs under xarray_parcel_amd/lib/; three alternating rounds, kernel time of the dominant kernel per run)
args=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done; shift
for rep in 1 2 3; do for L in "$@"; do
  export XPARCEL_LIB=$PWD/xarray_parcel_amd/lib/$L
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu --no-table-leg "${args[@]}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L kernel_ms %.4f' % d['roofline']['kernel_ms'])"
done; done
