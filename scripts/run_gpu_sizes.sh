# kernel time vs grid size (tail / quantisation effects): NY rows of 1024 columns, 64 levels fp64
for ny in ${*:-768 1024 3072 12288}; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --ny $ny 2>/dev/null > gpurun_out/size_$ny.log
  python - $ny <<'PY'
import sys, json
ny = sys.argv[1]
d = json.loads([l for l in open(f'gpurun_out/size_{ny}.log') if l.startswith('{')][-1])
ms = d['roofline']['kernel_ms']
print('NY', ny, 'col/s %.4g' % d['value'], 'kernel_ms %.4f' % ms, 'ns/column %.4f' % (ms * 1e6 / (int(ny) * 1024)))
PY
done
