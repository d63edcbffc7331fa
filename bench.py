#!/usr/bin/env python3
"""
bench.py -- headline benchmark (BASELINE.json): column-profiles/s for surface_based_cape_cin on
config c2, synthetic 64-level x 1024 x 1024 fp64 soundings, CAPE/CIN only, per GPU.

A "step" is one pass of the hot path (xp_cape_cin, surface parcel, exact moist mode, defaults of the
reference: virtual-temperature correction on, log LCL interpolation) over the rank's 1024 x 1024
columns, inputs already resident in HBM.  With N > 1 ranks (torchrun, one process per GPU) every rank
owns its own 1024 x 1024 slab of a (N*1024) x 1024 grid (weak scaling, no data-path collective) and the
per-column CAPE/CIN are gathered to rank 0 with one RCCL gather per step on a side stream.

Prints ONE JSON line on rank 0.  `roofline` is computed from HIP-event timings of the kernel launches
inside the timed region; `cpu_baseline` times the CPU oracle (C restatement, OpenMP, all host cores) on
a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NLEV, NY, NX = 64, 1024, 1024
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes_per_column(nlev, itemsize, n_out=2):
    # read p, T, Td once, write CAPE and CIN once
    return 3 * nlev * itemsize + n_out * itemsize


def measured_traffic(nlev, ny, nx, dtype):
    """HBM bytes per launch from the rocprofv3 PMC pass committed under profiles/ (FETCH_SIZE / WRITE_SIZE,
    corrected as MI355X_MICROARCH.md prescribes; see the JSON's note).  bench.py cannot run the profiler on itself,
    so this is the number measured on the same command at the time the profile was taken; None for any other shape."""
    import glob
    if (nlev, ny, nx, dtype) != (NLEV, NY, NX, 'f64'):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))['hbm_traffic_bytes'])
    except Exception:
        return None


def measured_valu_busy(nlev, ny, nx, dtype):
    """Fraction of SIMD cycles spent issuing VALU instructions, from the same committed PMC pass
    (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)): the kernel is fp64-VALU-bound, so this, not the
    HBM fraction, says how close it runs to the machine.  None for any other shape."""
    import glob
    if (nlev, ny, nx, dtype) != (NLEV, NY, NX, 'f64'):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json')))
    try:
        d = json.load(open(files[-1]))['per_launch_mean']
        return float(d['SQ_ACTIVE_INST_VALU'] * 4.0 / (d['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0))
    except Exception:
        return None


def cpu_baseline(seed, nlev, sample_cols):
    import numpy as np
    from oracle import c_oracle
    from xarray_parcel_amd import synth
    c_oracle.build()
    p, t, td = synth.columns(nlev=nlev, ncol=sample_cols, seed=seed, dtype=np.float64)
    c_oracle.cape_cin_grid(p[:, :2048], t[:, :2048], td[:, :2048], moist='rk4')          # warm up threads
    times = []
    for _ in range(5):                                   # protocol of parcel_test.py:31-35: wall clock, repeats, median
        t0 = time.perf_counter()
        c_oracle.cape_cin_grid(p, t, td, moist='rk4')
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {'value': sample_cols / dt, 'unit': 'column-profiles/s', 'cores': c_oracle.max_threads(), 'kind': 'port',
            'sample': f'{sample_cols} columns x {nlev} levels of the same synthetic workload, oracle/c/xp_oracle.c '
                      f'(OpenMP), median of 5 runs, {dt:.2f} s each'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--cpu-sample', type=int, default=0, help='columns for the CPU baseline (0 = auto)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--moist', default='exact', choices=['exact', 'family', 'table'],
                    help='exact = RK4 stepper (headline); family = same ODE from the adiabat-family table (xparcel.h); '
                         'table = the reference\'s lookup tables (pf.py:525-607), generated on the GPU before the timed region')
    ap.add_argument('--humidity', default='dewpoint', choices=['dewpoint', 'specific'],
                    help="'specific': feed specific humidity and convert on load (XP_HUM_SPECIFIC); not the headline")
    ap.add_argument('--data', default='hashed', choices=['hashed', 'smooth'],
                    help="'smooth': spatially correlated columns (sensitivity run; the headline uses SURVEY 8d's hashed columns)")
    ap.add_argument('--nlev', type=int, default=NLEV)
    ap.add_argument('--ny', type=int, default=NY)
    ap.add_argument('--nx', type=int, default=NX)
    a = ap.parse_args()

    # the in-tree library normally travels with the snapshot; if it is missing, local rank 0 builds it and the others wait
    from xarray_parcel_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        if int(os.environ.get('LOCAL_RANK', '0')) == 0:
            _lib.build()
        else:
            while not os.path.exists(_lib.LIB_PATH):
                time.sleep(1.0)
            time.sleep(2.0)

    import torch
    import torch.distributed as dist
    from xarray_parcel_amd import numpy_api as xa
    from xarray_parcel_amd import synth

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'
    # Rehearsal knobs (not used by the driver): XPARCEL_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # XPARCEL_BENCH_BACKEND=gloo gathers through host memory, so the N > 1 control flow can be exercised on a 1-GPU box.
    single = os.environ.get('XPARCEL_BENCH_SINGLE_DEVICE') == '1'
    backend = os.environ.get('XPARCEL_BENCH_BACKEND', 'nccl')
    local = 0 if single else local
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == 'nccl' else torch.device('cpu')       # where the gather buffers live

    tdt = torch.float64 if a.dtype == 'f64' else torch.float32
    ncol = a.ny * a.nx
    p, t, td = synth.columns_torch(a.nlev, ncol, dev, seed=20250719, dtype=tdt, col_offset=rank * ncol,
                                   smooth=(a.data == 'smooth'), nx=a.nx)
    if a.humidity == 'specific':                                   # q of air with the synthetic dewpoint (exact inversion)
        e = 6.112 * torch.exp(17.67 * (td - 273.15) / (td - 29.65))
        w = 0.6219569100577033 * e / (p - e)
        td = (w / (1.0 + w)).to(tdt)
        del e, w
    if a.moist == 'table':
        from xarray_parcel_amd import adiabat_tables
        adiabat_tables.load_moist_adiabat_lookups(cache=False)
    want = ('cape', 'cin')
    side = torch.cuda.Stream(device=dev) if world > 1 else None
    gathered = [torch.empty((world, 2, ncol), dtype=tdt, device=cdev) for _ in range(2)] if (world > 1 and rank == 0) else None
    sendbuf = [torch.empty((2, ncol), dtype=tdt, device=dev) for _ in range(2)] if world > 1 else None
    kernel_ms = []

    def step(i, timed):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = xa.cape_cin_columns(p, t, td, want=want, moist=a.moist, humidity=a.humidity)
        e1.record()
        if timed:
            kernel_ms.append((e0, e1))
        if world > 1:
            b = i & 1
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                sendbuf[b][0].copy_(r['cape'])
                sendbuf[b][1].copy_(r['cin'])
                src = sendbuf[b]
                if backend != 'nccl':
                    side.synchronize()
                    src = src.cpu()
                dist.gather(src, list(gathered[b].unbind(0)) if rank == 0 else None, dst=0)
            r['cape'].record_stream(side)
            r['cin'].record_stream(side)
        return r

    def fence():
        if world > 1:
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i, False)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        last = step(i, True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        kms = [e0.elapsed_time(e1) for e0, e1 in kernel_ms]
        avg_ms = sum(kms) / len(kms)
        item = 8 if a.dtype == 'f64' else 4
        bytes_launch = algorithmic_bytes_per_column(a.nlev, item) * ncol
        achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
        out = {
            'metric': 'column-profiles/sec for surface_based_cape_cin', 'value': world * ncol * a.steps / dt,
            'unit': 'column-profiles/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': a.dtype, 'data': 'synthetic' if a.data == 'hashed' else 'synthetic (spatially smooth variant, not the headline data)',
            'config': {'workload': f'c2: synthetic {a.nlev}-level x {a.ny} x {a.nx} {a.dtype} soundings per GPU, '
                                   f'surface_based_cape_cin (CAPE/CIN only), exact moist mode ({a.moist}), inputs resident in HBM',
                       'columns_per_gpu': ncol, 'levels': a.nlev,
                       'multi_gpu': 'y-slab per rank + one RCCL gather of (cape, cin) per step' if world > 1 else 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': measured_traffic(a.nlev, a.ny, a.nx, a.dtype),
                         'kernel': 'xp::k_cape_cin<%s, 0, false, %d, %s>' % ('double' if a.dtype == 'f64' else 'float',
                                                                            {'exact': 0, 'table': 1, 'family': 2}[a.moist],
                                                                            'true' if a.humidity == 'specific' else 'false'),
                         'kernel_ms': avg_ms, 'algorithmic_bytes_per_launch': bytes_launch,
                         'valu_busy': measured_valu_busy(a.nlev, a.ny, a.nx, a.dtype)},
            'check': {'max_cape': float(last['cape'].max()), 'min_cin': float(last['cin'].min())},
        }
        if not a.no_cpu and world == 1:
            try:
                sample = a.cpu_sample or (1 << 20)
                out['cpu_baseline'] = cpu_baseline(20250719, a.nlev, sample)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out['cpu_baseline'] = {'value': None, 'unit': 'column-profiles/s', 'cores': 0, 'kind': 'port',
                                       'sample': f'failed: {e}'}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
