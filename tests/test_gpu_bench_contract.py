"""bench.py prints ONE JSON line with the keys the driver reads (task contract) -- checked on a small grid."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
                          '--ny', '64', '--cpu-sample', '4096'], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['dtype'] == 'f64' and d['scaling'] == 'weak' and d['vs_baseline'] is None and 'workload' in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['traffic'] is None                      # PMC traffic is only quoted for the c2 shape it was measured on
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert d['value'] > 1e6


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """The N > 1 control flow of bench.py (slab per rank, side-stream gather, barrier + max-over-ranks timing) with two
    ranks sharing cuda:0 and a gloo gather -- the RCCL path proper needs the driver's multi-GPU node."""
    env = dict(os.environ, XPARCEL_BENCH_SINGLE_DEVICE='1', XPARCEL_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--ny', '64',
           '--no-cpu']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0 and 'cpu_baseline' not in d
    assert d['config']['columns_per_gpu'] == 64 * 1024
