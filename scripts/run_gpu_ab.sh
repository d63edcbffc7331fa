# A/B of library builds: run_gpu_ab.sh <bench args> -- lib1.so lib2.so ...   (libs under xarray_parcel_amd/lib/)
args=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done; shift
for L in "$@"; do
  export XPARCEL_LIB=$PWD/xarray_parcel_amd/lib/$L
  echo "== $L"
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu "${args[@]}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'])"
  bash scripts/run_gpu_pmc_mode.sh "${args[@]}" 2>&1 | grep -E "KERNEL.*k_cape_cin|INSTS_VALU|VMEM_RD|WAIT_ANY|WAVE_CYCLES|INSTS_LDS|WRITE_SIZE" | head -8
done
