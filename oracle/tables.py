"""
oracle/tables.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

The reference's moist-adiabat lookup tables and their use, restated:
  * build_tables()      pf.py:447-523 moist_adiabat_lookup (generation + painting rules)
  * moist_lapse_table() pf.py:525-607 moist_lapse (nearest-neighbour index, np.interp in linear p, masks)

The original cache files (adiabat_lookups/*.nc) are git-ignored upstream and absent from /root/reference, so
the tables are regenerated; the reference solves each adiabat with MetPy's LSODA (atol 1e-7, rtol 1.5e-8,
~15 ms each, 4 min in all), here all 14 300 are solved as one vector system with DOP853 (rtol = atol = 1e-9).
Cells on bin edges can therefore differ from the author's files: table-mode parity is defined against THESE
tables (the HIP path is handed the same arrays), and pinned to the reference only by the four 2-decimal
test_moist_lapse* KATs and the 0.037 K accuracy figure of parcel_functions_demo.ipynb:252.
Storage: index as uint16 (0 = NaN), adiabats as float32 (1.5e-5 K rounding, below LSODA's own error).
"""
import os
from dataclasses import dataclass

import numpy as np
from scipy.integrate import solve_ivp

from . import thermo as th

P_MAX, P_MIN, P_STEP = 1100.0, 2.5, 0.5            # pf.py:447-448: np.round(np.arange(1100, 2, -0.5), 1)
T_MIN, T_MAX, T_STEP = 173.0, 316.0, 0.02          # pf.py:449-450: np.round(np.arange(173, 316, 0.02), 2)


@dataclass
class Tables:
    index: np.ndarray      # uint16 [n_p][n_t]; pressure DEscending (1100 ... 2.5), 0 = NaN
    adiabats: np.ndarray   # float32 [n_adiabat][n_p]; pressure AScending (2.5 ... 1100) as after pf.py:54
    p_max: float = P_MAX
    p_step: float = P_STEP
    t_min: float = T_MIN
    t_step: float = T_STEP

    n_p = property(lambda self: self.index.shape[0])
    n_t = property(lambda self: self.index.shape[1])
    n_adiabat = property(lambda self: self.adiabats.shape[0])


def grids():
    pressure_levels = np.round(np.arange(1100, 2, step=-0.5), 1)
    temperatures = np.round(np.arange(173, 316, step=0.02), 2)
    return pressure_levels, temperatures


def round_to(x, to, dp=2):
    """pf.py:358-362."""
    return np.round(np.round(x / to) * to, dp)


def solve_adiabats(start_temperatures, pressure_levels, rtol=1e-9, atol=1e-9):
    """All adiabats from 1100 hPa at once: rows = adiabats, columns = pressure_levels (descending)."""
    t0 = np.asarray(start_temperatures, dtype=np.float64)
    res = solve_ivp(lambda p, t: th._moist_dt_dp(p, t), (pressure_levels[0], pressure_levels[-1]), t0,
                    method='DOP853', rtol=rtol, atol=atol, t_eval=pressure_levels)
    assert res.success
    return res.y


def paint(index, i, profile, pressure_levels, temperatures):
    """The two painting rules of pf.py:484-504 for adiabat number i."""
    n_p, n_t = index.shape
    t0_idx = int(round(temperatures[0] / T_STEP))
    p0 = pressure_levels[0]
    # (i) at every table pressure: cell (p, round(T_adiabat))
    jt = np.round(profile / T_STEP).astype(np.int64) - t0_idx
    ok = (jt >= 0) & (jt < n_t)
    index[np.nonzero(ok)[0], jt[ok]] = i
    # (ii) at every table temperature: cell (round(p_adiabat(T)), T); p by np.interp on the reversed curve
    ppt = np.interp(temperatures, profile[::-1], pressure_levels[::-1], left=np.nan, right=np.nan)
    ip = np.round((p0 - round_to(ppt, P_STEP)) / P_STEP)
    ok = ~np.isnan(ip) & (ip >= 0) & (ip < n_p)
    index[ip[ok].astype(np.int64), np.nonzero(ok)[0]] = i


def build_tables(verbose=False, adiabat_dtype=np.float32):
    pressure_levels, temperatures = grids()
    starts = np.empty(2 * len(temperatures))
    starts[0::2] = temperatures                          # offsets 0 and temp_step/2 (pf.py:479)
    starts[1::2] = temperatures + T_STEP / 2
    prof = solve_adiabats(starts, pressure_levels)       # [14300][2196], pressure descending
    index = np.zeros((len(pressure_levels), len(temperatures)), dtype=np.uint16)
    for i in range(prof.shape[0]):                       # later adiabats overwrite earlier ones
        paint(index, i + 1, prof[i], pressure_levels, temperatures)
    adiabats = np.ascontiguousarray(prof[:, ::-1].astype(adiabat_dtype))   # sortby('pressure') (pf.py:54)
    return Tables(index=index, adiabats=adiabats)


import tempfile
_CACHE = os.path.join(tempfile.gettempdir(), 'xparcel_oracle_adiabat_tables_v1.npz')   # 157 MB: kept out of the tree
_mem = {}


def get_tables(cache=True):
    """Build (about 10 s) or load the oracle's tables."""
    if 'tab' in _mem:
        return _mem['tab']
    if cache and os.path.exists(_CACHE):
        z = np.load(_CACHE)
        tab = Tables(index=z['index'], adiabats=z['adiabats'])
    else:
        tab = build_tables()
        if cache:
            os.makedirs(os.path.dirname(_CACHE), exist_ok=True)
            np.savez(_CACHE, index=tab.index, adiabats=tab.adiabats)
    _mem['tab'] = tab
    return tab


def nearest_index_descending(value, first, step, n):
    """pandas Index.get_indexer(method='nearest') on a monotonically DEcreasing index first, first-step, ...:
    the left (larger-value) neighbour wins ties (op = <= for non-increasing indexes)."""
    f = (first - value) / step
    i0 = np.floor(f)
    i = np.where((i0 + 1) - f < f - i0, i0 + 1, i0)
    return int(min(max(i, 0), n - 1))


def nearest_index_ascending(value, first, step, n):
    """Same on an increasing index: the right (larger-value) neighbour wins ties (op = <)."""
    f = (value - first) / step
    j0 = np.floor(f)
    j = np.where((j0 + 1) - f <= f - j0, j0 + 1, j0)
    return int(min(max(j, 0), n - 1))


def moist_lapse_table(tab, pressure, parcel_temperature, parcel_pressure):
    """pf.py:525-607 for one parcel: returns temperatures at `pressure` (1-D)."""
    p = np.asarray(pressure, dtype=np.float64)
    out = np.full(p.shape, np.nan)
    if tab is None:
        raise AssertionError('Call load_moist_adiabat_lookups first.')       # pf.py:60
    if np.isnan(parcel_temperature) or np.isnan(parcel_pressure):
        return out
    ip = nearest_index_descending(parcel_pressure, tab.p_max, tab.p_step, tab.n_p)
    jt = nearest_index_ascending(parcel_temperature, tab.t_min, tab.t_step, tab.n_t)
    a = int(tab.index[ip, jt])
    if a == 0:
        return out                                                            # NaN cell (pf.py:570-582)
    row = tab.adiabats[a - 1].astype(np.float64)
    p_min = tab.p_max - (tab.n_p - 1) * tab.p_step
    xp = p_min + tab.p_step * np.arange(tab.n_p)
    ok = ~np.isnan(p) & (p >= p_min) & (p <= tab.p_max)                       # pf.py:598-605
    out[ok] = np.interp(p[ok], xp, row)
    return out
