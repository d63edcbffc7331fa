"""
Moist-adiabat lookup tables in the reference's format (pf.py:39-61, 318-356, 447-523).

The reference builds them with 14 300 serial MetPy ODE solves (minutes) and caches two NetCDF files under
./adiabat_lookups/ (git-ignored upstream, so not available).  Here the 14 300 adiabats x 2 196 pressures are
integrated on the GPU by the library's own exact moist mode (xp_moist_lapse, RK4 with steps <= 0.1 in ln p,
well under a second) and painted into the index table on the host with the reference's rules; the result is
cached as one .npz and handed to the library with xp_set_tables.

Storage differs from the reference (uint16 index with 0 = NaN instead of float64/NaN; float32 adiabats):
values are identical up to float32 rounding of the curves (1.5e-5 K, below the reference's LSODA tolerance).
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L

P_MAX, P_STEP, T_MIN, T_STEP = 1100.0, 0.5, 173.0, 0.02
_state = {'tables': None}


def _grids():
    pressure_levels = np.round(np.arange(1100, 2, step=-0.5), 1)           # pf.py:447-448
    temperatures = np.round(np.arange(173, 316, step=0.02), 2)             # pf.py:449-450
    return pressure_levels, temperatures


def _round_to(x, to, dp=2):
    return np.round(np.round(x / to) * to, dp)                             # pf.py:358-362


def moist_adiabat_lookup():
    """pf.py:447-523: returns (index uint16 [2196][7150] over descending pressure, adiabats float32
    [14300][2196] over ASCENDING pressure)."""
    from . import numpy_api
    pressure_levels, temperatures = _grids()
    starts = np.empty(2 * len(temperatures))
    starts[0::2] = temperatures                                            # offsets 0, temp_step/2 (pf.py:479)
    starts[1::2] = temperatures + T_STEP / 2
    pgrid = np.repeat(pressure_levels[:, None], len(starts), axis=1)       # (2196, 14300): one column per adiabat
    prof = np.asarray(numpy_api.moist_lapse(pgrid, starts, None, moist='exact')).T   # reference pressure = 1100 hPa
    n_p, n_t = len(pressure_levels), len(temperatures)
    index = np.zeros((n_p, n_t), dtype=np.uint16)
    t0_idx = int(round(temperatures[0] / T_STEP))
    for i in range(prof.shape[0]):                                         # "last writer wins" (pf.py:488, 503)
        pr = prof[i]
        jt = np.round(pr / T_STEP).astype(np.int64) - t0_idx               # pf.py:484-489
        ok = (jt >= 0) & (jt < n_t)
        index[np.nonzero(ok)[0], jt[ok]] = i + 1
        ppt = np.interp(temperatures, pr[::-1], pressure_levels[::-1], left=np.nan, right=np.nan)   # pf.py:495-497
        ip = np.round((P_MAX - _round_to(ppt, P_STEP)) / P_STEP)           # pf.py:499-504
        ok = ~np.isnan(ip) & (ip >= 0) & (ip < n_p)
        index[ip[ok].astype(np.int64), np.nonzero(ok)[0]] = i + 1
    return index, np.ascontiguousarray(prof[:, ::-1].astype(np.float32))


def set_tables(index, adiabats):
    """Hand tables (any origin, reference layout) to the library."""
    index = np.ascontiguousarray(index, dtype=np.uint16)
    adiabats = np.ascontiguousarray(adiabats, dtype=np.float32)
    assert adiabats.shape[1] == index.shape[0], 'adiabats must be [n_adiabat][n_pressure]'
    assert int(index.max()) <= adiabats.shape[0], 'index table refers to an adiabat that is not there'
    lib = L.init()
    t = L.Tables(index.shape[0], index.shape[1], adiabats.shape[0], P_MAX, P_STEP, T_MIN, T_STEP,
                 index.ctypes.data, adiabats.ctypes.data)
    L.check(lib.xp_set_tables(C.byref(t)))
    _state['tables'] = (index, adiabats)


def default_cache_path(base_dir=None):
    base = base_dir or os.environ.get('XPARCEL_CACHE', os.path.join(os.path.expanduser('~'), '.cache', 'xparcel'))
    return os.path.join(base, 'adiabat_lookups', 'moist_adiabat_tables_v1.npz')


def moist_adiabat_tables(regenerate=False, cache=True, base_dir=None):
    """pf.py:318-356."""
    path = default_cache_path(base_dir)
    if not regenerate and os.path.exists(path):
        z = np.load(path)
        index, adiabats = z['index'], z['adiabats']
        # a stale or foreign cache file must not reach the device: shapes of the current grid, index entries in range
        n_p, n_t = len(np.arange(P_MAX, 2, -P_STEP)), len(np.round(np.arange(T_MIN, 316, T_STEP), 2))
        if (index.shape == (n_p, n_t) and adiabats.ndim == 2 and adiabats.shape[1] == n_p and
                int(index.max()) <= adiabats.shape[0]):
            return index, adiabats
    index, adiabats = moist_adiabat_lookup()
    if cache:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        np.savez(path, index=index, adiabats=adiabats)
    return index, adiabats


def load_moist_adiabat_lookups(**kwargs):
    """pf.py:39-61: load (or generate and cache) the tables and make them resident on the device."""
    index, adiabats = moist_adiabat_tables(**kwargs)
    set_tables(index, adiabats)


def tables():
    return _state['tables']
