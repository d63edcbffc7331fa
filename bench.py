#!/usr/bin/env python3
"""
bench.py -- headline benchmark (BASELINE.json): column-profiles/s for surface_based_cape_cin, per config.

  --config c2 (default, the headline): synthetic 64-level x 1024 x 1024 fp64 soundings PER GPU, CAPE/CIN only.  With N > 1
            ranks every rank owns its own slab of a (N*1024) x 1024 grid (weak scaling).
  --config c4: ONE fixed 128-level x 8192 x 8192 fp32 grid, surface-based CAPE/CIN, cut into N contiguous y-slabs
            (strong scaling: the 1/2/4/8-GPU curve of SURVEY.md 8e).
  --config c5: ONE fixed 24-timestep x 100-level x 2048 x 2048 fp32 grid, most-unstable AND mixed-layer CAPE/CIN per
            step, the flattened (time, y) axis cut into N slabs (strong scaling).

A "step" is one pass of the hot path (xp_cape_cin through the C ABI; reference defaults: virtual-temperature
correction on, log LCL interpolation) over the rank's columns, inputs already resident in HBM.  Columns are independent:
no data-path collective; the per-column CAPE/CIN are gathered to rank 0 with one RCCL gather per step (per parcel) on a
side stream.  (The reference's counterpart: dask chunks over the horizontal dims, pf.py:343-346, parcel_test.py:604.)

Prints ONE JSON line on rank 0.  `roofline` is computed from HIP-event timings of the dominant kernel's launches inside
the timed region; `cpu_baseline` times the CPU oracle (C restatement, OpenMP, all host cores) on a bounded sample of the
same workload; `table_mode` / `rk4_mode` (c2, one GPU) are the same step in the reference's shipping moist mode (its
lookup tables, pf.py:525-607) and with the RK4 stepper, measured after the headline's timed region.

Legs measured after the headline (never a reason to lose it: each one is guarded):
  one GPU, --config c2:  "c3" (BASELINE config 3 at full size: 128 x 4096 x 4096 f32, full profile + scalars), "c4_share" and
            "c5_share" (one of eight ranks' slab of configs 4 / 5; c5 = most-unstable + mixed-layer, as two passes and as the
            fused pass of xp_cape_cin_multi), each with kernel_ms, frac on SURVEY 8(d)'s algorithmic bytes and a strided
            >= 4000-column check against the oracle; "cpu_baseline_numpy" (the one-column NumPy restatement, the closest
            analogue of the reference's xarray / NumPy path, on <= 2048 columns).  --leg-scale N divides their column counts.
  any N, --config c2 (the driver's command line):  "strong_c4": the FIXED 128 x 8192 x 8192 f32 grid of config 4 cut into N
            y-slabs (SURVEY 8e's strong-scaling curve; N = 1: the whole 103 GB grid on the one GPU), same step / gather /
            timing protocol as the headline.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (5.1-5.8 TB/s is what copy / triad kernels reach)
MOIST_ID = {'exact': 0, 'table': 1, 'family': 2}
PARCEL_ID = {'surface': 0, 'most_unstable': 1, 'mixed_layer': 2}
CONFIGS = {
    'c2': dict(nlev=64, nt=1, ny=1024, nx=1024, dtype='f64', parcels=('surface',), scaling='weak', seed=20250719,
               what='synthetic 64-level x 1024 x 1024 f64 soundings per GPU'),
    'c4': dict(nlev=128, nt=1, ny=8192, nx=8192, dtype='f32', parcels=('surface',), scaling='strong', seed=20250721,
               what='one synthetic 128-level x 8192 x 8192 f32 grid cut into y-slabs'),
    'c5': dict(nlev=100, nt=24, ny=2048, nx=2048, dtype='f32', parcels=('most_unstable', 'mixed_layer'), scaling='strong',
               seed=20250722, what='one synthetic 24-timestep x 100-level x 2048 x 2048 f32 grid, (time, y) cut into slabs'),
}


def algorithmic_bytes_per_column(nlev, itemsize, n_out=2):
    # read p, T, Td once, write CAPE and CIN once
    return 3 * nlev * itemsize + n_out * itemsize


def persist_min_cols(parcel):     # csrc/xparcel.hip xp_cape_cin: family mode runs persistent wavefronts on grids this large
    e = os.environ.get('XP_PERSIST_MIN_COLS')
    return int(e) if e is not None else (1 << 19) if parcel in ('most_unstable', 'mixed_layer') else (3 << 18)


def kernel_name(dtype, parcel, moist, humidity, ncol):
    # <T, parcel mode, profile output, moist mode, specific-humidity input, default options, CAPE/CIN-only outputs,
    # persistent wavefronts>; the dispatch rule of csrc/xp_cape_tu.hip: the default-options + CAPE/CIN-only instantiation for
    # what bench.py asks for; persistent wavefronts for family mode on large grids
    hum = humidity == 'specific'
    spec = not hum                    # bench.py asks for CAPE / CIN only with the default options: DEF + LEAN in every moist mode
    tf = lambda b: 'true' if b else 'false'
    return 'xp::k_cape_cin<%s, %d, false, %d, %s, %s, %s, %s>' % ('double' if dtype == 'f64' else 'float', PARCEL_ID[parcel], MOIST_ID[moist],
                                                                   tf(hum), tf(spec), tf(spec), tf(moist == 'family' and ncol >= persist_min_cols(parcel)))


def profile_counters(kernel, shape):
    """HBM traffic / VALU-busy of `kernel` from a rocprofv3 PMC pass committed under profiles/ (bench.py cannot run the
    profiler on itself) -- quoted ONLY when that pass was taken on these very kernel sources (csrc fingerprint) and on
    this shape; otherwise the fields stay null and `source.stale` says why.  FETCH_SIZE / WRITE_SIZE are corrected as
    MI355X_MICROARCH.md prescribes (the JSON's note)."""
    from xarray_parcel_amd import _lib
    sha = _lib.csrc_sha()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc.json'))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get('kernel') == kernel and d.get('shape') == list(shape):
            best = (f, d)
    if best is None:
        return None, None, {'file': None, 'csrc_sha': sha, 'stale': True, 'why': 'no committed PMC pass for this kernel / shape'}
    f, d = best
    src = {'file': os.path.relpath(f, ROOT), 'csrc_sha': d.get('csrc_sha'), 'current_csrc_sha': sha, 'stale': d.get('csrc_sha') != sha}
    if src['stale']:
        src['why'] = 'kernel sources changed since the PMC pass'
        return None, None, src
    m = d['per_launch_mean']
    valu_busy = float(m['SQ_ACTIVE_INST_VALU'] * 4.0 / (m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0))
    return float(d['hbm_traffic_bytes']), valu_busy, src


def cpu_baseline(seed, nlev, sample_cols, parcel, moist):
    import numpy as np
    from oracle import c_oracle
    from xarray_parcel_amd import synth
    c_oracle.build()
    p, t, td = synth.columns(nlev=nlev, ncol=sample_cols, seed=seed, dtype=np.float64)
    omode = 'family' if moist == 'family' else 'rk4'        # (table mode: the oracle's tables are not built here; RK4 stands in)
    c_oracle.cape_cin_grid(p[:, :2048], t[:, :2048], td[:, :2048], moist=omode, parcel=parcel)   # warm up threads, build tables
    times = []
    for _ in range(9):                                   # protocol of parcel_test.py:31-35: wall clock, repeats, median (~10-12 s of CPU work in all)
        t0 = time.perf_counter()
        c_oracle.cape_cin_grid(p, t, td, moist=omode, parcel=parcel)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {'value': sample_cols / dt, 'unit': 'column-profiles/s', 'cores': c_oracle.max_threads(), 'kind': 'port',
            'sample': f'{sample_cols} columns x {nlev} levels of the same synthetic workload ({parcel} parcel), '
                      f'oracle/c/xp_oracle.c (OpenMP, moist mode {omode}), median of 9 runs, {dt:.2f} s each'}


def _event_ms(fn, steps=5, warmup=2):
    """Mean HIP-event duration of `fn` (one launch sequence on torch's current stream) over `steps` calls."""
    import torch
    r = None
    for _ in range(warmup):
        r = fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(steps):
        del r
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / len(ev), r


def _oracle_check(res, p, t, td, parcel, depth, moist, min_cols=4000, profile=False):
    """Strided sample of >= min_cols columns against the CPU oracle (the checker, not the thing measured): LFC / EL /
    parcel indices must be identical, CAPE / CIN are reported as max abs differences (f32 outputs: their rounding)."""
    import numpy as np
    import torch
    from oracle import c_oracle
    ncol = p.shape[1]
    stride = max(1, ncol // min_cols - 1) | 1
    idx = torch.arange(0, ncol, stride, device=p.device)
    kw = {} if depth is None else {'depth': depth}
    ref = c_oracle.cape_cin_grid(p[:, idx].cpu().numpy(), t[:, idx].cpu().numpy(), td[:, idx].cpu().numpy(), parcel=parcel,
                                 moist='family' if moist == 'family' else 'rk4', want_profile=profile, **kw)
    out = {'columns': int(idx.numel()), 'stride': int(stride)}
    keys = [k for k in ('lfc_index', 'el_index', 'parcel_index') if k in res]
    if keys:
        out['indices_identical'] = bool(all(np.array_equal(res[k][idx].cpu().numpy(), ref[k]) for k in keys))
    for k in ('cape', 'cin'):
        out[k + '_maxdiff'] = float(np.max(np.abs(res[k][idx].cpu().numpy().astype(np.float64) - ref[k])))
    if profile:
        a_ = res['profile']['temperature'][:, idx].cpu().numpy().astype(np.float64)
        b_ = ref['profile']['temperature']
        out['profile_nan_pattern_identical'] = bool(np.array_equal(np.isnan(a_), np.isnan(b_)))
        out['profile_T_maxdiff'] = float(np.nanmax(np.abs(a_ - b_)))
    return out


def config_legs(a, dev, moist):
    """BASELINE configs 3, 4 (one of eight slabs) and 5 (one of eight slabs) on this GPU at their configured sizes
    (divided by --leg-scale), after the headline: kernel time by HIP events, roofline fraction on SURVEY 8(d)'s algorithmic
    bytes, strided oracle check.  The reference's counterpart is benchmark_cape (parcel_test.py:586-619)."""
    import torch
    from xarray_parcel_amd import numpy_api as xa
    from xarray_parcel_amd import synth
    legs = {}
    sc = max(1, a.leg_scale)
    want_i = ('cape', 'cin', 'lfc_index', 'el_index', 'parcel_index')

    def frac(bytes_, ms):
        return bytes_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS

    # ---- c3: 128 x 4096 x 4096 f32, full profile + LCL / LFC / EL + CAPE / CIN ------------------------------------------
    try:
        nlev, ncol = 128, 4096 * 4096 // sc
        p, t, td = synth.columns_torch(nlev, ncol, dev, seed=20250720, dtype=torch.float32)
        alg = (3 * nlev * 4 + 6 * (nlev + 1) * 4 + 13 * 4) * ncol
        leg = {'what': f'BASELINE config 3: {nlev} x {ncol} columns f32, full profile (6 x {nlev + 1} rows) + all scalars', 'columns': ncol,
               'algorithmic_bytes_per_launch': alg}
        for m in dict.fromkeys((moist, 'exact')):
            ms, r = _event_ms(lambda: xa.cape_cin_columns(p, t, td, want_profile=True, moist=m), steps=3, warmup=1)
            leg[m] = {'kernel_ms': ms, 'value': ncol / ms * 1e3, 'achieved': alg / (ms * 1e-3) / 1e9, 'frac': frac(alg, ms),
                      'check': _oracle_check(r, p, t, td, 'surface', None, m, profile=True)}
            del r
        legs['c3'] = leg
        del p, t, td
    except Exception as e:
        legs['c3'] = {'error': str(e)}
    torch.cuda.empty_cache()
    # ---- c4, one of eight y-slabs: 128 x (1024 x 8192) f32, surface-based CAPE / CIN ---------------------------------------
    try:
        nlev, ncol = 128, 8192 * 8192 // 8 // sc
        p, t, td = synth.columns_torch(nlev, ncol, dev, seed=20250721, dtype=torch.float32)
        alg = algorithmic_bytes_per_column(nlev, 4) * ncol
        ms, r = _event_ms(lambda: xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist=moist))
        full = xa.cape_cin_columns(p, t, td, want=want_i, moist=moist)
        legs['c4_share'] = {'what': f'BASELINE config 4, one of 8 ranks: {nlev} x {ncol} columns f32, surface_based_cape_cin (CAPE/CIN only)',
                            'columns': ncol, 'moist': moist, 'kernel_ms': ms, 'value': ncol / ms * 1e3, 'algorithmic_bytes_per_launch': alg,
                            'achieved': alg / (ms * 1e-3) / 1e9, 'frac': frac(alg, ms), 'gather_payload_MB_per_rank': 2 * 4 * ncol / 1e6,
                            'check': _oracle_check(full, p, t, td, 'surface', None, moist)}
        del p, t, td, r, full
    except Exception as e:
        legs['c4_share'] = {'error': str(e)}
    torch.cuda.empty_cache()
    # ---- c5, one of eight (time, y)-slabs: 100 x (3 x 2048 x 2048) f32, most-unstable + mixed-layer --------------------------
    try:
        nlev, ncol = 100, 24 * 2048 * 2048 // 8 // sc
        p, t, td = synth.columns_torch(nlev, ncol, dev, seed=20250722, dtype=torch.float32)
        alg = algorithmic_bytes_per_column(nlev, 4) * ncol
        leg = {'what': f'BASELINE config 5, one of 8 ranks: {nlev} x {ncol} columns f32, most_unstable_cape_cin (300 hPa) + '
                       f'mixed_layer_cape_cin (100 hPa), CAPE/CIN only; frac on the bytes of ONE pass over the grid',
               'columns': ncol, 'moist': moist, 'algorithmic_bytes_per_launch': alg}
        step_ms = 0.0
        for name, depth in (('most_unstable', 300.0), ('mixed_layer', 100.0)):
            ms, r = _event_ms(lambda: xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, want=('cape', 'cin'), moist=moist))
            full = xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, want=want_i, moist=moist)
            leg[name] = {'kernel_ms': ms, 'value': ncol / ms * 1e3, 'achieved': alg / (ms * 1e-3) / 1e9, 'frac': frac(alg, ms),
                         'check': _oracle_check(full, p, t, td, name, depth, moist)}
            step_ms += ms
            del r, full
        leg['step_ms'] = step_ms
        leg['step_value'] = ncol / step_ms * 1e3
        if moist == 'family':
            pcs = [('most_unstable', 300.0), ('mixed_layer', 100.0)]
            ms, r = _event_ms(lambda: xa.cape_cin_multi(p, t, td, pcs, want=('cape', 'cin'), moist=moist, fused=True))
            sep = [xa.cape_cin_columns(p, t, td, parcel=n_, depth=d_, want=('cape', 'cin'), moist=moist) for n_, d_ in pcs]
            same = all(bool(torch.equal(g[k], s_[k]) or torch.equal(torch.isnan(g[k]), torch.isnan(s_[k]))) for g, s_ in zip(r, sep) for k in ('cape', 'cin'))
            leg['fused_step'] = {'what': 'the same two parcels in ONE pass (xp_cape_cin_multi, XP_OPT_FUSE_PARCELS): opt-in, not the default',
                                 'kernel_ms': ms, 'value': ncol / ms * 1e3, 'frac': frac(alg, ms), 'bitwise_equal_to_two_passes': bool(same)}
            del r, sep
        legs['c5_share'] = leg
        del p, t, td
    except Exception as e:
        legs['c5_share'] = {'error': str(e)}
    torch.cuda.empty_cache()
    return legs


def cpu_baseline_numpy(seed, nlev, ncol=2048):
    """The one-column NumPy restatement of the reference (oracle/parcel_oracle.py: pf.py's array semantics op for op,
    MetPy's formulas restated, RK4 moist lapse), one process: the closest analogue available here of the reference's own
    xarray / NumPy path, which cannot be imported.  Protocol of parcel_test.py:31-35."""
    import numpy as np
    from oracle import parcel_oracle as po
    from xarray_parcel_amd import synth
    p, t, td = synth.columns(nlev=nlev, ncol=ncol, seed=seed, dtype=np.float64)
    po.set_moist_lapse('rk4')
    import warnings
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cape = [float(po.surface_based_cape_cin(p[:, c], t[:, c], td[:, c])[0]['cape']) for c in range(ncol)]
    dt = time.perf_counter() - t0
    return {'value': ncol / dt, 'unit': 'column-profiles/s', 'cores': 1, 'kind': 'port',
            'sample': f'{ncol} columns x {nlev} levels of the same synthetic workload, oracle/parcel_oracle.py (NumPy, one column at a time, '
                      f'one process), {dt:.1f} s; reference xarray path as published: ~2.8e3 column-profiles/s on 10 201 x 90 (BASELINE.md)',
            'max_cape': max(cape)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--cpu-sample', type=int, default=0, help='columns for the CPU baseline (0 = auto)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-table-leg', action='store_true', help='skip the table-mode leg of the c2 / one-GPU run')
    ap.add_argument('--no-config-legs', action='store_true', help='skip the c3 / c4_share / c5_share / numpy / strong_c4 legs')
    ap.add_argument('--preroll-seconds', type=float, default=0.5,
                    help='untimed run of the same step before the W warmup + K timed steps, so that they see steady-state clocks '
                         '(0: measure from an idle GPU; the first W + K steps are reported as cold_start either way)')
    ap.add_argument('--leg-scale', type=int, default=1, help='divide the column counts of the config legs by this (rehearsals, tests)')
    ap.add_argument('--dtype', default=None, choices=['f64', 'f32'])
    ap.add_argument('--moist', default='family', choices=['exact', 'family', 'table'],
                    help='family (headline) = the pseudo-adiabat ODE served from the adiabat-family table (xparcel.h: within 7.5e-7 K '
                         'of the ODE, all 59 reference KATs pass through the C ABI in this mode); exact = the same ODE by the RK4 '
                         'stepper (2e-5 K); table = the reference\'s lookup tables (pf.py:525-607), generated on the GPU before '
                         'the timed region')
    ap.add_argument('--humidity', default='dewpoint', choices=['dewpoint', 'specific'],
                    help="'specific': feed specific humidity and convert on load (XP_HUM_SPECIFIC); not the headline")
    ap.add_argument('--data', default='hashed', choices=['hashed', 'smooth'],
                    help="'smooth': spatially correlated columns (sensitivity run; the headline uses SURVEY 8d's hashed columns)")
    ap.add_argument('--nlev', type=int, default=None)
    ap.add_argument('--nt', type=int, default=None)
    ap.add_argument('--ny', type=int, default=None, help='override the grid (rehearsals on reduced sizes)')
    ap.add_argument('--nx', type=int, default=None)
    a = ap.parse_args()
    cfg = dict(CONFIGS[a.config])
    for k in ('nlev', 'nt', 'ny', 'nx', 'dtype'):
        if getattr(a, k) is not None:
            cfg[k] = getattr(a, k)
    reduced = any(cfg[k] != CONFIGS[a.config][k] for k in ('nlev', 'nt', 'ny', 'nx', 'dtype'))

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
    from xarray_parcel_amd.distributed import slab_bounds

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

    def run_cfg(cfg, moist, steps, warmup):
        """One workload on this rank's slab: data in HBM, `warmup` untimed steps, then exactly `steps` steps between
        barrier + synchronize fences, wall time = max over ranks."""
        tdt = torch.float64 if cfg['dtype'] == 'f64' else torch.float32
        nlev, nx = cfg['nlev'], cfg['nx']
        rows = cfg['nt'] * cfg['ny']                               # the sharded axis: y, or the flattened (time, y)
        if cfg['scaling'] == 'weak':                               # every rank its own grid of the configured size
            r0, r1, total_rows = rank * rows, (rank + 1) * rows, rows * world
        else:                                                      # one fixed grid cut into `world` slabs
            r0, r1 = slab_bounds(rows, world, rank)
            total_rows = rows
        ncol = (r1 - r0) * nx
        ncol_max = (max(slab_bounds(rows, world, r)[1] - slab_bounds(rows, world, r)[0] for r in range(world)) * nx
                    if cfg['scaling'] == 'strong' else ncol)
        # (generated directly on the owning GPU, 1 Mi columns at a time: no second copy of a 100 GB grid)
        p, t, td = synth.columns_torch(nlev, ncol, dev, seed=cfg['seed'], dtype=tdt, col_offset=r0 * nx,
                                       smooth=(a.data == 'smooth'), nx=nx)
        if a.humidity == 'specific':                               # q of air with the synthetic dewpoint (exact inversion)
            e = 6.112 * torch.exp(17.67 * (td - 273.15) / (td - 29.65))
            w = 0.6219569100577033 * e / (p - e)
            td = (w / (1.0 + w)).to(tdt)
            del e, w
        if moist == 'table':
            from xarray_parcel_amd import adiabat_tables
            adiabat_tables.load_moist_adiabat_lookups(cache=False)
        want = ('cape', 'cin')
        parcels = cfg['parcels']
        side = torch.cuda.Stream(device=dev) if world > 1 else None
        npar = len(parcels)
        gathered = [torch.empty((world, 2 * npar, ncol_max), dtype=tdt, device=cdev) for _ in range(2)] if (world > 1 and rank == 0) else None
        sendbuf = [torch.zeros((2 * npar, ncol_max), dtype=tdt, device=dev) for _ in range(2)] if world > 1 else None
        kernel_ms = {pc: [] for pc in parcels}

        def step(i, timed, moist):
            res = {}
            for pc in parcels:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                res[pc] = xa.cape_cin_columns(p, t, td, parcel=pc, want=want, moist=moist, humidity=a.humidity)
                e1.record()
                if timed is not None:
                    timed[pc].append((e0, e1))
            if world > 1:
                b = i & 1
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for j, pc in enumerate(parcels):
                        sendbuf[b][2 * j, :ncol].copy_(res[pc]['cape'])
                        sendbuf[b][2 * j + 1, :ncol].copy_(res[pc]['cin'])
                    src = sendbuf[b]
                    if backend != 'nccl':
                        side.synchronize()
                        src = src.cpu()
                    dist.gather(src, list(gathered[b].unbind(0)) if rank == 0 else None, dst=0)   # the ONE collective per step
                for pc in parcels:
                    res[pc]['cape'].record_stream(side)
                    res[pc]['cin'].record_stream(side)
            return res

        def fence():
            if world > 1:
                torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        def timed_steps(moist, steps, warmup, timed):
            for i in range(warmup):
                step(i, None, moist)
            fence()
            t0 = time.perf_counter()
            last = None
            for i in range(steps):
                last = step(i, timed, moist)
            fence()
            dt = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            return dt, last

        dt, last = timed_steps(moist, steps, warmup, kernel_ms)
        return dict(dt=dt, last=last, kernel_ms=kernel_ms, ncol=ncol, total_rows=total_rows, nlev=nlev, nx=nx, parcels=parcels,
                    timed_steps=timed_steps, tdt=tdt)

    def strong_leg(c4):
        """SURVEY 8(e)'s fixed-grid point for this N: config 4's ONE grid cut into `world` y-slabs, the headline's protocol."""
        nst = max(2, min(a.steps, 5))
        s4 = run_cfg(c4, a.moist, nst, 1)
        k4 = sum(e0.elapsed_time(e1) for e0, e1 in s4['kernel_ms']['surface']) / len(s4['kernel_ms']['surface'])
        leg = {'what': f"BASELINE config 4 (strong scaling): ONE fixed {c4['nlev']}-level x {c4['ny']} x {c4['nx']} f32 grid cut into {world} y-slab(s), "
                       f"surface_based_cape_cin{' + one gather per step' if world > 1 else ''}; same timing protocol as the headline",
               'scaling': 'strong', 'n_gpus': world, 'steps': nst, 'columns_total': s4['total_rows'] * s4['nx'], 'columns_this_rank': s4['ncol'],
               'value': s4['total_rows'] * s4['nx'] * nst / s4['dt'], 'unit': 'column-profiles/s', 'ms_per_step': s4['dt'] / nst * 1e3,
               'kernel_ms_rank0': k4, 'frac_rank0': algorithmic_bytes_per_column(c4['nlev'], 4) * s4['ncol'] / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
               'check': {'max_cape': float(s4['last']['surface']['cape'].max())}}
        s4 = None
        torch.cuda.empty_cache()
        return leg

    h = run_cfg(cfg, a.moist, a.steps, a.warmup)
    # Steady state.  A run that starts on an idle GPU measures its clock ramp: on MI355X the first ~10 ms of this kernel run
    # ~7 % slower than the same kernel half a second later (and ~15 % slower than the boost the card reaches in between:
    # DESIGN.md 7 "Clocks").  The driver's command line (--steps 20 --warmup 5) is 14 ms of GPU time in all.  So: the W + K
    # steps just measured are reported as `cold_start`, then the same step runs untimed for --preroll-seconds, and W untimed
    # + exactly K timed steps are taken again for `value`.  (Same number of extra steps on every rank: derived from the
    # all-reduced cold time.)
    cold = None
    if a.preroll_seconds > 0:
        ck = h['kernel_ms']
        cold = {'ms_per_step': h['dt'] / a.steps * 1e3,
                'kernel_ms_by_parcel': {pc: sum(e0.elapsed_time(e1) for e0, e1 in v) / len(v) for pc, v in ck.items()}}
        n_pre = max(1, int(a.preroll_seconds / max(h['dt'] / a.steps, 1e-6)))
        h['timed_steps'](a.moist, 0, n_pre, None)
        k2 = {pc: [] for pc in h['parcels']}
        dt2, last2 = h['timed_steps'](a.moist, a.steps, a.warmup, k2)
        h['dt'], h['last'], h['kernel_ms'] = dt2, last2, k2
        cold['preroll_steps'] = n_pre
    dt, last, kernel_ms, ncol, total_rows = h['dt'], h['last'], h['kernel_ms'], h['ncol'], h['total_rows']
    nlev, nx, parcels = h['nlev'], h['nx'], h['parcels']

    # N > 1 with the driver's command line (no --config): the fixed-grid curve of SURVEY 8(e) as a leg -- config 4's ONE
    # 128 x 8192 x 8192 f32 grid cut into N y-slabs, every rank generating its own slab in place
    strong = None
    if world > 1 and a.config == 'c2' and not a.no_config_legs:
        c4 = dict(CONFIGS['c4'])
        c4['ny'] = max(world, c4['ny'] // max(1, a.leg_scale))
        head_keep = {pc: {'max_cape': float(last[pc]['cape'].max()), 'min_cin': float(last[pc]['cin'].min())} for pc in parcels}
        per_parcel_head = {pc: sum(e0.elapsed_time(e1) for e0, e1 in v) / len(v) for pc, v in kernel_ms.items()}
        h = last = None                                            # the headline's grid makes room
        torch.cuda.empty_cache()
        strong = strong_leg(c4)
    if rank == 0:
        item = 8 if cfg['dtype'] == 'f64' else 4
        per_parcel = per_parcel_head if strong is not None else {pc: sum(e0.elapsed_time(e1) for e0, e1 in v) / len(v) for pc, v in kernel_ms.items()}
        dom = max(per_parcel, key=per_parcel.get)                  # the dominant kernel of the step
        avg_ms = per_parcel[dom]
        bytes_launch = algorithmic_bytes_per_column(nlev, item) * ncol
        achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
        kname = kernel_name(cfg['dtype'], dom, a.moist, a.humidity, ncol)
        traffic, valu_busy, src = profile_counters(kname, (nlev, ncol))
        total_cols = total_rows * nx
        parcel_txt = ' + '.join({'surface': 'surface_based_cape_cin', 'most_unstable': 'most_unstable_cape_cin',
                                 'mixed_layer': 'mixed_layer_cape_cin'}[pc] for pc in parcels)
        out = {
            'metric': 'column-profiles/sec for surface_based_cape_cin' if parcels == ('surface',) else
                      'column-profiles/sec for most_unstable_cape_cin + mixed_layer_cape_cin (one column-profile = both parcels)',
            'value': total_cols * a.steps / dt,
            'unit': 'column-profiles/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': cfg['scaling'], 'vs_baseline': None,
            'dtype': cfg['dtype'], 'data': 'synthetic' if a.data == 'hashed' else 'synthetic (spatially smooth variant, not the headline data)',
            'config': {'workload': f"{a.config}{' (REDUCED grid: rehearsal, not the configured size)' if reduced else ''}: {cfg['what']}, "
                                   f"{nlev} levels x {cfg['nt']} x {cfg['ny']} x {nx}, {parcel_txt} (CAPE/CIN only), "
                                   f"moist mode {a.moist}, inputs resident in HBM",
                       'columns_total': total_cols, 'columns_this_rank': ncol, 'levels': nlev,
                       'multi_gpu': (f"{'own grid per rank' if cfg['scaling'] == 'weak' else 'y-slab of the one grid per rank'} + one RCCL "
                                     f"gather of (cape, cin) per step") if world > 1 else 'single GPU',
                       'rccl_path': 'exercised in this run' if (world > 1 and backend == 'nccl') else
                                    ('not exercised (gloo rehearsal)' if world > 1 else 'n/a (one GPU)')},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': src,
                         'kernel': kname, 'kernel_ms': avg_ms, 'algorithmic_bytes_per_launch': bytes_launch,
                         'valu_busy': valu_busy,
                         'kernel_ms_by_parcel': per_parcel},
            'check': head_keep if strong is not None else {pc: {'max_cape': float(last[pc]['cape'].max()), 'min_cin': float(last[pc]['cin'].min())} for pc in parcels},
        }
        if cold is not None:
            ck = cold['kernel_ms_by_parcel'][dom]
            out['preroll'] = {'seconds': a.preroll_seconds, 'untimed_steps': cold['preroll_steps'], 'before': 'the W untimed + K timed steps of `value`', 'after': 'the W + K steps of `cold_start`'}
            out['cold_start'] = {'what': f"the first {a.warmup} + {a.steps} steps of this process, from an idle GPU (clock ramp), before the "
                                         f"{cold['preroll_steps']} untimed steps (~{a.preroll_seconds} s) that precede the steps `value` is taken from",
                                 'ms_per_step': cold['ms_per_step'], 'value': total_cols / (cold['ms_per_step'] * 1e-3), 'kernel_ms': ck,
                                 'frac': bytes_launch / (ck * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if strong is not None:
            out['strong_c4'] = strong
        if a.config == 'c2' and world == 1 and not a.no_table_leg and a.humidity == 'dewpoint':
            # the same step in the other two moist modes, measured after the headline's timed region: the reference's
            # shipping mode (its lookup tables) and the RK4 stepper
            legs = {'table': ('table_mode', "same step with the reference's lookup-table moist mode (pf.py:525-607; 31 MB index + 126 MB "
                                            "adiabat tables resident in HBM, regenerated here)"),
                    'exact': ('rk4_mode', 'same step with the RK4 stepper (XP_MOIST_EXACT)'),
                    'family': ('family_mode', 'same step with the adiabat-family table (XP_MOIST_FAMILY)')}
            for mode, (key, what) in legs.items():
                if mode == a.moist:
                    continue
                try:
                    t_gen = None
                    if mode == 'table':
                        from xarray_parcel_amd import adiabat_tables
                        t1 = time.perf_counter()
                        adiabat_tables.load_moist_adiabat_lookups(cache=False)      # generated on the GPU, outside any timed region
                        t_gen = time.perf_counter() - t1
                    tk = {pc: [] for pc in parcels}
                    dtt, _ = h['timed_steps'](mode, a.steps, 2, tk)
                    tms = sum(e0.elapsed_time(e1) for e0, e1 in tk[dom]) / len(tk[dom])
                    tname = kernel_name(cfg['dtype'], dom, mode, a.humidity, ncol)
                    ttraffic, _, tsrc = profile_counters(tname, (nlev, ncol))
                    out[key] = {'what': what, 'value': total_cols * a.steps / dtt, 'ms_per_step': dtt / a.steps * 1e3, 'kernel': tname,
                                'kernel_ms': tms, 'achieved': bytes_launch / (tms * 1e-3) / 1e9,
                                'frac': bytes_launch / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': ttraffic, 'traffic_source': tsrc}
                    if t_gen is not None:
                        out[key]['table_generation_s'] = t_gen
                except Exception as e:  # a report next to the headline, never a reason to lose it
                    out[key] = {'error': str(e)}
        if not a.no_cpu and world == 1:
            try:
                sample = a.cpu_sample or ((1 << 20) if nlev <= 64 else (1 << 19))
                out['cpu_baseline'] = cpu_baseline(cfg['seed'], nlev, sample, dom, a.moist)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out['cpu_baseline'] = {'value': None, 'unit': 'column-profiles/s', 'cores': 0, 'kind': 'port',
                                       'sample': f'failed: {e}'}
        if a.config == 'c2' and world == 1 and not a.no_config_legs and a.humidity == 'dewpoint':
            h = last = None
            torch.cuda.empty_cache()
            try:
                out.update(config_legs(a, dev, a.moist))
            except Exception as e:
                out['config_legs_error'] = str(e)
            try:                                            # the N = 1 point of the fixed-grid curve: the whole 103 GB grid on this GPU
                c4 = dict(CONFIGS['c4'])
                c4['ny'] = max(1, c4['ny'] // max(1, a.leg_scale))
                out['strong_c4'] = strong_leg(c4)
            except Exception as e:
                out['strong_c4'] = {'error': str(e)}
                torch.cuda.empty_cache()
            if not a.no_cpu:
                try:
                    out['cpu_baseline_numpy'] = cpu_baseline_numpy(cfg['seed'], nlev, 2048 if not a.cpu_sample else max(32, min(2048, a.cpu_sample // 16)))
                except Exception as e:
                    out['cpu_baseline_numpy'] = {'value': None, 'sample': f'failed: {e}'}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
