"""
oracle/family.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

Specification of the build's second exact moist mode, the "adiabat family": instead of integrating MetPy's
pseudo-adiabat ODE level by level (oracle/thermo.py moist_lapse_rk4), the one-parameter family of its solutions
T(x ; psi), x = ln p, psi = the adiabat's temperature at 1000 hPa, is stored once -- as the parcel's VIRTUAL temperature
along the adiabat, Tv(x ; psi) = T (1 + 0.608 w_s(p, T)) (pf.py:760-775: what the CAPE / CIN integration consumes) -- in
a piecewise polynomial:

    x-pieces   j = 0..NPX-1 : [XHI - WX*(j+1), XHI - WX*j],  XHI = ln 1100, WX = 0.5  (1100 hPa ... ~20.1 hPa)
    psi-pieces q = 0..NPS-1 : [EDGES[q], EDGES[q+1]]  (215 ... 312 K, narrower towards the warm end)
    Tv(x ; psi) = sum_n sum_m A[j][n][m][q] z^n s^m,   z, s = the piece-local coordinates in [-1, 1],  n, m <= 8

A[j][.][.][q] is the interpolant at the 9 x 9 Chebyshev nodes of the piece (the ODE by RK4 from ln 1000 with steps
<= 1/80 in ln p, error ~1e-10 K), converted to monomials.  A column's label psi is found by one coarse RK4 march from
its LCL to 1000 hPa (steps <= 0.25) followed by three Newton steps on the table inside the psi-piece the coarse label
falls in, so that the tabulated curve passes through (p_lcl, Tv_lcl).  The parcel TEMPERATURE, where it is wanted
(profile output, no virtual-temperature correction), is the T that has this virtual temperature at this pressure:
Tv = T (1 + 0.608 w_s(p, T)) solved by Newton -- so T and Tv stay tied by the reference's own formula to rounding.
Above the table's top (~20 hPa) the curve continues dry (~ p^kappa; e_s / p < 1e-6 there).  This plays the role of the
reference's own lookup tables (pf.py:447-607) but is accurate to < 1e-6 K against the ODE instead of 0.037 K
(tests/test_oracle_family.py) -- closer to the ODE than the RK4 stepper (2e-5 K) or MetPy's LSODA tolerance; parcels
whose label or LCL lie outside the table fall back to the RK4 mode.

The product builds the same table in xp_init by the same recipe, written independently; the parity tests hand the
oracle's table to the device so that both evaluate identical numbers.
"""
import numpy as np

from . import thermo as th

XHI = float(np.log(1100.0))
WX = 0.5
NPX = 8
XLO = XHI - WX * NPX            # ~ ln 20.15 hPa: table top
NDEG = 8                        # degree in x (9 coefficients per piece)
MDEG = 8                        # degree in psi
EDGES = np.array([215.0, 245.0, 262.0, 275.0, 285.0, 293.0, 299.0, 304.0, 308.5, 312.0])
NPS = len(EDGES) - 1
X1000 = float(np.log(1000.0))
BUILD_H = 0.0125                # RK4 step bound of the table build
LABEL_H = 0.25                  # RK4 step bound of the coarse label
LABEL_MARGIN = 0.05             # coarse label must be this far inside [EDGES[0], EDGES[-1]]; Newton may move it this far
NEWTON_STEPS = 3


def _f(x, t):
    return th._moist_dt_dlnp(x, t)


def _rk4(x, t, x1, h_max):
    """March from x to x1 in ceil(|x1 - x| / h_max) equal RK4 steps (t may be a vector)."""
    n = max(1, int(np.ceil(abs(x1 - x) / h_max - 1e-12)))
    h = (x1 - x) / n
    for _ in range(n):
        k1 = _f(x, t)
        k2 = _f(x + 0.5 * h, t + 0.5 * h * k1)
        k3 = _f(x + 0.5 * h, t + 0.5 * h * k2)
        k4 = _f(x + h, t + h * k3)
        t = t + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        x = x + h
    return t


def virtual_temperature(p, t):
    """Tv of saturated air at (p, T): pf.py:760 + 775."""
    return t * (1.0 + th.VT_EPSILON * th.saturation_mixing_ratio(p, t))


def temperature_of(p, tv, steps=5):
    """The T with virtual_temperature(p, T) = tv: Newton from T0 = tv / (1 + 0.608 w_s(p, tv)), a fixed number of
    steps (1e-13 K after five wherever e_s <= 0.1 p, i.e. everywhere on the table; 4e-8 K after four).  (The device's
    profile kernels reach the same root from a warm start -- Tv - T of the node before -- in three steps, with a residual
    test that adds two: same value to 1e-13 K, csrc/xp_device.hpp Family::temperature_from.)"""
    c = th.VT_EPSILON * th.EPSILON
    t = tv / (1.0 + th.VT_EPSILON * th.saturation_mixing_ratio(p, tv))
    for _ in range(steps):
        e = th.saturation_vapor_pressure(t)
        de = e * (17.67 * 243.5) / ((t - 29.65) * (t - 29.65))
        f = t * (1.0 + c * e / (p - e)) - tv
        df = 1.0 + c * e / (p - e) + t * c * p * de / ((p - e) * (p - e))
        t = t - f / df
    return t


def x_mid(j):
    return XHI - WX * (j + 0.5)


def psi_mid(q):
    return 0.5 * (EDGES[q] + EDGES[q + 1])


def psi_half(q):
    return 0.5 * (EDGES[q + 1] - EDGES[q])


def build_table():
    """(NPX, NDEG+1, MDEG+1, NPS) float64.  Nodes: Chebyshev points of the first kind of every piece; every psi-node's
    adiabat is marched away from ln 1000 on both sides through the x-nodes in order of distance."""
    un = np.cos(np.pi * (np.arange(NDEG + 1) + 0.5) / (NDEG + 1))
    um = np.cos(np.pi * (np.arange(MDEG + 1) + 0.5) / (MDEG + 1))
    xs = np.concatenate([x_mid(j) + 0.5 * WX * un for j in range(NPX)])
    ps = np.concatenate([psi_mid(q) + psi_half(q) * um for q in range(NPS)])
    vals = np.empty((xs.size, ps.size))
    for side in (xs <= X1000, xs > X1000):
        idx = np.nonzero(side)[0]
        idx = idx[np.argsort(np.abs(xs[idx] - X1000), kind='stable')]
        x, t = X1000, ps.copy()
        for i in idx:
            if xs[i] != x:
                t = _rk4(x, t, xs[i], BUILD_H)
                x = xs[i]
            vals[i] = virtual_temperature(np.exp(xs[i]), t)
    vn = np.vander(un, NDEG + 1, increasing=True)
    vm = np.vander(um, MDEG + 1, increasing=True)
    tab = np.empty((NPX, NDEG + 1, MDEG + 1, NPS))
    for j in range(NPX):
        for q in range(NPS):
            blk = vals[j * (NDEG + 1):(j + 1) * (NDEG + 1), q * (MDEG + 1):(q + 1) * (MDEG + 1)]
            a = np.linalg.solve(vn, blk)                # monomials in z (rows) ...
            tab[j, :, :, q] = np.linalg.solve(vm, a.T).T   # ... and in s (columns)
    return tab


_cache = {}


def table():
    if 'tab' not in _cache:
        _cache['tab'] = build_table()
    return _cache['tab']


def x_piece(x):
    """Piece of ln p = x: floor((XHI - x) / WX) clipped to the table; x must lie in [XLO, XHI]."""
    return int(min(max(np.floor((XHI - x) * (1.0 / WX)), 0), NPX - 1))


def psi_piece(psi):
    return int(min(max(np.searchsorted(EDGES, psi, side='right') - 1, 0), NPS - 1))


def _horner(c, u):
    v = c[-1]
    for k in range(len(c) - 2, -1, -1):
        v = v * u + c[k]
    return v


def column_poly(tab, j, q, s):
    """Coefficients in z of x-piece j for a column with psi-coordinate s in psi-piece q: c_n = sum_m A[j][n][m][q] s^m."""
    return np.array([_horner(tab[j, n, :, q], s) for n in range(NDEG + 1)])


def label(tab, x_lcl, t_lcl, p_lcl=None):
    """(psi, q) with Tv(x_lcl ; psi) = Tv_lcl on the table, or (nan, -1) when the parcel is outside it.  p_lcl: the
    pressure whose logarithm x_lcl is, when the caller has it (exp(x_lcl) otherwise)."""
    if not (np.isfinite(x_lcl) and np.isfinite(t_lcl)) or not (XLO <= x_lcl <= XHI):
        return np.nan, -1
    psi0 = float(_rk4(x_lcl, t_lcl, X1000, LABEL_H)) if x_lcl != X1000 else float(t_lcl)
    if not (EDGES[0] + LABEL_MARGIN <= psi0 <= EDGES[-1] - LABEL_MARGIN):
        return np.nan, -1
    q = psi_piece(psi0)
    j = x_piece(x_lcl)
    tv_lcl = float(virtual_temperature(np.exp(x_lcl) if p_lcl is None else p_lcl, t_lcl))
    z = (x_lcl - x_mid(j)) * (2.0 / WX)
    b = np.array([_horner(tab[j, :, m, q], z) for m in range(MDEG + 1)])      # polynomial in s at x = x_lcl
    db = b[1:] * np.arange(1, MDEG + 1)
    inv_h = 1.0 / psi_half(q)
    psi = psi0
    for _ in range(NEWTON_STEPS):
        s = (psi - psi_mid(q)) * inv_h
        psi = psi - (_horner(b, s) - tv_lcl) / (_horner(db, s) * inv_h)
    if not (abs(psi - psi0) <= LABEL_MARGIN):
        return np.nan, -1
    return psi, q


def evaluate_tv(tab, x, psi, q):
    """Tv(x ; psi) for a label found by label(); NaN for x > XHI; dry continuation above the table top."""
    if not np.isfinite(x) or x > XHI:
        return np.nan
    s = (psi - psi_mid(q)) * (1.0 / psi_half(q))
    if x < XLO:
        t_top = _horner(column_poly(tab, NPX - 1, q, s), -1.0)
        return float(t_top * np.exp(th.KAPPA * (x - XLO)))
    j = x_piece(x)
    z = (x - x_mid(j)) * (2.0 / WX)
    return float(_horner(column_poly(tab, j, q, s), z))


def evaluate(tab, x, psi, q):
    """The parcel temperature at ln p = x: the T whose virtual temperature is the tabulated one."""
    tv = evaluate_tv(tab, x, psi, q)
    return float(temperature_of(np.exp(x), tv)) if np.isfinite(tv) else np.nan


def moist_lapse_family(pressure, parcel_temperature, parcel_pressure=None):
    """Family-mode moist_lapse with the RK4 fallback, same calling convention as thermo.moist_lapse_rk4."""
    p = np.atleast_1d(np.asarray(pressure, dtype=np.float64))
    if parcel_pressure is None:
        parcel_pressure = p[0]
    tab = table()
    psi, q = label(tab, np.log(parcel_pressure), parcel_temperature, parcel_pressure) if parcel_pressure > 0 else (np.nan, -1)
    if not np.isnan(psi):
        with np.errstate(invalid='ignore', divide='ignore'):
            x = np.log(p)
        ok = ~np.isnan(p)
        vals = np.array([(float(temperature_of(pp, evaluate_tv(tab, xx, psi, q))) if xx <= XHI else np.nan) if o else np.nan
                         for xx, pp, o in zip(x, p, ok)])
        vals[p == parcel_pressure] = parcel_temperature
        if not np.any(ok & np.isnan(vals)):
            return vals
    # the label or a level (p > 1100 hPa) outside the table: the whole parcel takes the RK4 mode, as the device fix-up
    # pass does
    return th.moist_lapse_rk4(p, parcel_temperature, parcel_pressure)
