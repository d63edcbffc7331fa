timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -2 gpurun_out/pytest_gpu.log; grep AssertionError gpurun_out/pytest_gpu.log | cut -c1-300; timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/bench.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1]); print('BENCH', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --moist family > gpurun_out/bench_family.log 2>&1; python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_family.log') if l.startswith('{')][-1]); print('BENCH family', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
