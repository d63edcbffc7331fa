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
                          '--ny', '64', '--cpu-sample', '4096', '--leg-scale', '256'], capture_output=True, text=True, timeout=900, cwd=ROOT)
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
    # PMC traffic is only quoted for the shape AND the kernel sources it was measured on; otherwise null + the reason
    assert r['traffic'] is None and r['traffic_source']['stale'] is True
    assert 'table_mode' in d and d['table_mode'].get('kernel_ms', 0) > 0, d.get('table_mode')   # the reference's shipping mode
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c
    assert d['value'] > 1e6
    # the steps `value` comes from follow a steady-state pre-roll; the W + K steps from the idle GPU are reported beside them
    cs = d['cold_start']
    assert cs['kernel_ms'] > 0 and cs['ms_per_step'] > 0 and 0 < cs['frac'] < 1 and cs['value'] > 1e6
    # the other BASELINE configs as legs on the driver's clock (reduced by --leg-scale here), each with its oracle check
    for k in ('c3', 'c4_share', 'c5_share'):
        assert k in d and 'error' not in d[k], (k, d.get(k))
    for m in ('family', 'exact'):
        c3 = d['c3'][m]
        assert c3['kernel_ms'] > 0 and 0 < c3['frac'] < 1 and c3['check']['indices_identical'] and c3['check']['profile_nan_pattern_identical']
        assert c3['check']['columns'] >= 4000 and c3['check']['cape_maxdiff'] < 5e-3 and c3['check']['profile_T_maxdiff'] < 1e-4
    assert d['c4_share']['kernel_ms'] > 0 and d['c4_share']['check']['indices_identical'] and d['c4_share']['check']['columns'] >= 4000
    c5 = d['c5_share']
    for pc in ('most_unstable', 'mixed_layer'):
        assert c5[pc]['kernel_ms'] > 0 and c5[pc]['check']['indices_identical'] and c5[pc]['check']['cape_maxdiff'] < 5e-3
    assert abs(c5['step_ms'] - c5['most_unstable']['kernel_ms'] - c5['mixed_layer']['kernel_ms']) < 1e-9
    assert c5['fused_step']['bitwise_equal_to_two_passes'] is True and c5['fused_step']['kernel_ms'] > 0
    n = d['cpu_baseline_numpy']
    assert n['kind'] == 'port' and n['cores'] == 1 and n['value'] > 0
    s4 = d['strong_c4']                                    # the N = 1 point of SURVEY 8(e)'s fixed-grid curve
    assert s4['scaling'] == 'strong' and s4['n_gpus'] == 1 and s4['columns_total'] == (8192 // 256) * 8192 and s4['value'] > 0


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """The N > 1 control flow of bench.py (slab per rank, side-stream gather, barrier + max-over-ranks timing) with two
    ranks sharing cuda:0 and a gloo gather -- the RCCL path proper needs the driver's multi-GPU node."""
    env = dict(os.environ, XPARCEL_BENCH_SINGLE_DEVICE='1', XPARCEL_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29517', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--ny', '64',
           '--no-cpu', '--leg-scale', '64']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['value'] > 0 and 'cpu_baseline' not in d
    assert d['config']['columns_this_rank'] == 64 * 1024 and d['config']['columns_total'] == 2 * 64 * 1024
    # the fixed-grid (strong-scaling) curve of SURVEY 8(e) rides along with the driver's command line as a leg
    s4 = d['strong_c4']
    assert s4['scaling'] == 'strong' and s4['n_gpus'] == 2 and s4['columns_total'] == 128 * 8192 and s4['columns_this_rank'] == 64 * 8192
    assert s4['value'] > 0 and s4['check']['max_cape'] > 100.0


def test_bench_fixed_grid_configs_two_ranks_rehearsal():
    """--config c4 / c5: ONE fixed grid cut into y-slabs (strong scaling), two ranks on one GPU, uneven slabs; and the
    one-rank run of the same reduced grid gives the same check values (the slabs reproduce the whole)."""
    env = dict(os.environ, XPARCEL_BENCH_SINGLE_DEVICE='1', XPARCEL_BENCH_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    for cfgname, extra, rows in (('c4', ['--ny', '37', '--nx', '512'], 37), ('c5', ['--nt', '3', '--ny', '11', '--nx', '256'], 33)):
        res = {}
        for n in (1, 2):
            cmd = ([sys.executable] + (['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                                        '127.0.0.1', '--master-port', '29519'] if n == 2 else []) +
                   [os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '2', '--warmup', '1', '--config', cfgname, '--no-cpu'] + extra)
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
            assert out.returncode == 0, out.stderr[-2000:]
            lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
            assert len(lines) == 1
            res[n] = json.loads(lines[0])
        d1, d2 = res[1], res[2]
        assert d1['scaling'] == d2['scaling'] == 'strong' and d2['n_gpus'] == 2 and d1['dtype'] == 'f32'
        assert d1['config']['columns_total'] == d2['config']['columns_total']          # the same grid, whatever N
        assert d2['config']['columns_this_rank'] == (rows + 1) // 2 * int(extra[-1])   # rank 0 owns the larger slab
        assert 'REDUCED' in d1['config']['workload'] and cfgname in d1['config']['workload']
        if cfgname == 'c5':
            assert set(d1['roofline']['kernel_ms_by_parcel']) == {'most_unstable', 'mixed_layer'}
