for n in 16 32 64 128; do timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --nlev $n 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('NLEV', $n, 'kernel_ms', round(d['roofline']['kernel_ms'],4))"; done
