"""
oracle/family.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Specification of the build's second exact moist mode, the "adiabat family": instead of integrating MetPy's
pseudo-adiabat ODE level by level (oracle/thermo.py moist_lapse_rk4), the family of its solutions is tabulated
once,

    TAB[i][j] = T(X_i ; psi_j),   X_i = ln(30 hPa) + i*DX (i < NX),   psi_j = 215 K + j*0.5 K (j < NS),

where psi labels an adiabat by its temperature at 1000 hPa, and evaluated by 6 x 6 Lagrange interpolation
(6 nodes in ln p around the level, 6 in psi around the column's label).  A column's label is the root of
T(ln p_lcl ; psi) = T_lcl (Newton on the interpolant), so the interpolated adiabat passes through the LCL exactly.
This plays the role of the reference's own lookup tables (pf.py:447-607) but is accurate to 1.4e-6 K against the
ODE instead of 0.037 K (tests/test_oracle_family.py); points outside the table fall back to the RK4 mode.

The table is built by classical RK4 in ln p with step DX/8 from 1000 hPa outwards (error ~1e-11 K), the same
recipe the product's xp_init uses -- written independently on both sides; the parity tests hand the oracle's
table to the device so that both interpolate identical numbers.
"""
import numpy as np

from . import thermo as th

XLO = float(np.log(30.0))
DX = 0.028
NX = 133                      # up to ln(30) + 132*0.028 = ln(1208 hPa)
SLO, DS, NS = 215.0, 0.5, 201  # 215 ... 315 K
X1000 = float(np.log(1000.0))
SUB = 8                        # RK4 substeps per table interval


def _f(x, t):
    return th._moist_dt_dlnp(x, t)


def build_table():
    """(NX, NS) float64: vectorised over psi, RK4 with step DX/SUB marching away from ln 1000 on both sides.
    X_i are not aligned with ln 1000, so the first leg on each side is a partial step."""
    psi = SLO + DS * np.arange(NS)
    tab = np.empty((NX, NS))
    i_up = int(np.floor((X1000 - XLO) / DX))          # last node with X_i <= ln 1000
    for direction in (-1, +1):
        t = psi.copy()
        x = X1000
        rng = range(i_up, -1, -1) if direction < 0 else range(i_up + 1, NX)
        for i in rng:
            x1 = XLO + DX * i
            h = (x1 - x) / SUB
            for _ in range(SUB):
                k1 = _f(x, t)
                k2 = _f(x + 0.5 * h, t + 0.5 * h * k1)
                k3 = _f(x + 0.5 * h, t + 0.5 * h * k2)
                k4 = _f(x + h, t + h * k3)
                t = t + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
                x = x + h
            x = x1
            tab[i] = t
    return tab


_cache = {}


def table():
    if 'tab' not in _cache:
        _cache['tab'] = build_table()
    return _cache['tab']


def lagrange6(t):
    """Weights of the 6-point Lagrange interpolant on nodes -2..3 at fractional position t in [0, 1)."""
    a, b, c, d, e, g = t + 2.0, t + 1.0, t, t - 1.0, t - 2.0, t - 3.0
    return np.array([-(b * c * d * e * g) / 120.0, (a * c * d * e * g) / 24.0, -(a * b * d * e * g) / 12.0,
                     (a * b * c * e * g) / 12.0, -(a * b * c * d * g) / 24.0, (a * b * c * d * e) / 120.0])


def dlagrange6(t):
    """d/dt of lagrange6 (product rule on the five-factor numerators)."""
    n = np.array([-2.0, -1.0, 0.0, 1.0, 2.0, 3.0])
    den = np.array([-120.0, 24.0, -12.0, 12.0, -24.0, 120.0])
    out = np.zeros(6)
    for k in range(6):
        others = [m for m in range(6) if m != k]
        s = 0.0
        for skip in others:
            prod = 1.0
            for m in others:
                if m != skip:
                    prod *= (t - n[m])
            s += prod
        out[k] = s / den[k]
    return out


def in_x_range(x):
    i = np.floor((x - XLO) / DX)
    return (i - 2 >= 0) & (i + 3 <= NX - 1)


def in_psi_range(psi):
    j = np.floor((psi - SLO) / DS)
    return (j - 2 >= 0) & (j + 3 <= NS - 1)


def evaluate(tab, x, psi):
    """T(x ; psi) by 6 x 6 Lagrange interpolation; NaN outside the table's interior."""
    if not (np.isfinite(x) and np.isfinite(psi)) or not in_x_range(x) or not in_psi_range(psi):
        return np.nan
    ux = (x - XLO) / DX
    i = int(np.floor(ux))
    us = (psi - SLO) / DS
    j = int(np.floor(us))
    return float(lagrange6(ux - i) @ tab[i - 2:i + 4, j - 2:j + 4] @ lagrange6(us - j))


def label(tab, x_lcl, t_lcl, max_iter=12):
    """psi with T(x_lcl ; psi) = t_lcl: Newton from the first-order guess t_lcl + dT/dlnp * (ln 1000 - x_lcl), clamped
    to the table's interior; returns NaN when x_lcl is outside or the root sits on the clamp."""
    if not (np.isfinite(x_lcl) and np.isfinite(t_lcl)) or not in_x_range(x_lcl):
        return np.nan
    lo, hi = SLO + 2.0 * DS, SLO + DS * (NS - 3) - 1e-9
    psi = min(max(t_lcl + float(_f(x_lcl, t_lcl)) * (X1000 - x_lcl), lo), hi)
    ux = (x_lcl - XLO) / DX
    i = int(np.floor(ux))
    wx = lagrange6(ux - i)
    for _ in range(max_iter):
        us = (psi - SLO) / DS
        j = int(np.floor(us))
        col = wx @ tab[i - 2:i + 4, j - 2:j + 4]
        fval = float(col @ lagrange6(us - j)) - t_lcl
        dval = float(col @ dlagrange6(us - j)) / DS
        step = fval / dval
        new = min(max(psi - step, lo), hi)
        done = abs(new - psi) < 1e-10
        psi = new
        if done:
            break
    if abs(evaluate(tab, x_lcl, psi) - t_lcl) > 1e-8:
        return np.nan                                  # label outside the table: caller falls back to RK4
    return psi


def moist_lapse_family(pressure, parcel_temperature, parcel_pressure=None):
    """Family-mode moist_lapse with the RK4 fallback, same calling convention as thermo.moist_lapse_rk4."""
    p = np.atleast_1d(np.asarray(pressure, dtype=np.float64))
    if parcel_pressure is None:
        parcel_pressure = p[0]
    tab = table()
    psi = label(tab, np.log(parcel_pressure), parcel_temperature) if parcel_pressure > 0 else np.nan
    out = np.full(p.shape, np.nan)
    if not np.isnan(psi):
        with np.errstate(invalid='ignore', divide='ignore'):
            x = np.log(p)
        ok = ~np.isnan(p)
        vals = np.array([evaluate(tab, xx, psi) if o else np.nan for xx, o in zip(x, ok)])
        out = vals
        out[p == parcel_pressure] = parcel_temperature
        if not np.any(ok & np.isnan(vals)):
            return out
    # any level (or the label) outside the table: the whole parcel takes the RK4 mode, as the device fix-up pass does
    return th.moist_lapse_rk4(p, parcel_temperature, parcel_pressure)
