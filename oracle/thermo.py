"""
oracle/thermo.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

NumPy/SciPy restatement of the MetPy-1.4.1 thermodynamic formulas that the
reference (traupach/xarray_parcel, modules/parcel_functions.py, "pf.py") calls
but does not vendor.  MetPy is a third-party dependency absent from
/root/reference; the only pin is the notebook output "MetPy 1.4.1"
(parcel_functions_demo.ipynb:86).  The formulas below are MetPy's published
algorithms (Bolton 1980 etc.), anchored on the reference's call sites and pinned
by the reference's own known-answer tests (modules/unit_tests.py), see
tests/test_oracle_kat.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (xarray_parcel_amd/) never does.

Units: pressure hPa, temperature K, mixing ratio kg/kg.
"""
import numpy as np
from scipy import integrate as _si
from scipy import optimize as _so

# -- constants (metpy.constants, 1.4.1) ---------------------------------------
R_GAS = 8.314462618            # J / mol / K
MW_WATER = 18.015268           # g / mol
MW_DRY = 28.96546              # g / mol
RD = R_GAS / MW_DRY * 1000.0   # 287.04749097718457 J/kg/K  (pf.py:1361 mpconsts.Rd.m)
EPSILON = MW_WATER / MW_DRY    # 0.6219569100577033
KAPPA = 2.0 / 7.0              # mpconsts.kappa (pf.py:313)
CP_D = RD / KAPPA              # 1004.6662184201462
LV = 2.50084e6                 # J/kg
SAT_PRESSURE_0C = 6.112        # hPa
P0 = 1000.0                    # hPa
VT_EPSILON = 0.608             # hard-coded in pf.py:782


def saturation_vapor_pressure(temperature):
    """Bolton (1980) eq. 10; metpy.calc.saturation_vapor_pressure."""
    t = np.asarray(temperature, dtype=np.float64)
    return SAT_PRESSURE_0C * np.exp(17.67 * (t - 273.15) / (t - 29.65))


def dewpoint(vapor_pressure):
    """Inverse of Bolton's formula; metpy.calc.dewpoint (pf.py:280), result in K."""
    val = np.log(np.asarray(vapor_pressure, dtype=np.float64) / SAT_PRESSURE_0C)
    return 273.15 + 243.5 * val / (17.67 - val)


def mixing_ratio(partial_press, total_press):
    """metpy.calc.mixing_ratio: eps * e / (p - e)."""
    return EPSILON * partial_press / (total_press - partial_press)


def saturation_mixing_ratio(total_press, temperature):
    """metpy.calc.saturation_mixing_ratio (pf.py:258, pf.py:760)."""
    return mixing_ratio(saturation_vapor_pressure(temperature), total_press)


def vapor_pressure(pressure, mixing):
    """metpy.calc.vapor_pressure (pf.py:275)."""
    return pressure * mixing / (EPSILON + mixing)


def relative_humidity_from_dewpoint(temperature, dewpt):
    """metpy.calc.relative_humidity_from_dewpoint (pf.py:698)."""
    return saturation_vapor_pressure(dewpt) / saturation_vapor_pressure(temperature)


def mixing_ratio_from_relative_humidity(pressure, temperature, relative_humidity):
    """MetPy 1.4.1 form (pf.py:701): rh * w_s(p, T).  (1.6 changed this.)"""
    return relative_humidity * saturation_mixing_ratio(pressure, temperature)


def potential_temperature(pressure, temperature):
    """metpy.calc.potential_temperature (pf.py:253)."""
    return temperature / exner_function(pressure)


def exner_function(pressure):
    """metpy.calc.exner_function (pf.py:269)."""
    return (np.asarray(pressure, dtype=np.float64) / P0) ** KAPPA


def equivalent_potential_temperature(pressure, temperature, dewpt):
    """Bolton (1980) eq. 39; metpy.calc.equivalent_potential_temperature (pf.py:123)."""
    t = np.asarray(temperature, dtype=np.float64)
    td = np.asarray(dewpt, dtype=np.float64)
    e = saturation_vapor_pressure(td)
    r = saturation_mixing_ratio(pressure, td)
    t_l = 56.0 + 1.0 / (1.0 / (td - 56.0) + np.log(t / td) / 800.0)
    th_l = potential_temperature(pressure - e, t) * (t / t_l) ** (0.28 * r)
    return th_l * np.exp(r * (1.0 + 0.448 * r) * (3036.0 / t_l - 1.78))


def dewpoint_from_specific_humidity(pressure, temperature, specific_humidity):
    """MetPy 1.4.1 form used by the harness (parcel_test.py:262).  Not pinned by
    any KAT: parity unpinned for this one function."""
    w = specific_humidity / (1.0 - specific_humidity)
    rh = w / saturation_mixing_ratio(pressure, temperature)
    return dewpoint(rh * saturation_vapor_pressure(temperature))


# -- LCL -------------------------------------------------------------------------
def _lcl_iter(p, p0, w, t):
    td = dewpoint(vapor_pressure(p, w))
    return p0 * (td / t) ** (1.0 / KAPPA)


def lcl_metpy(pressure, temperature, dewpt, max_iters=50, eps=1e-5):
    """metpy.calc.lcl (1.4.1) on scalars/arrays: SciPy fixed_point (Steffensen),
    xtol 1e-5, then the isclose snap to the starting pressure.  Used at pf.py:644.
    With array input SciPy stops when *all* elements pass, exactly like MetPy."""
    pressure = np.asarray(pressure, dtype=np.float64)
    temperature = np.asarray(temperature, dtype=np.float64)
    dewpt = np.asarray(dewpt, dtype=np.float64)
    w = mixing_ratio(saturation_vapor_pressure(dewpt), pressure)
    lcl_p = _so.fixed_point(_lcl_iter, pressure, args=(pressure, w, temperature),
                            xtol=eps, maxiter=max_iters)
    lcl_p = np.where(np.isclose(lcl_p, pressure), pressure, lcl_p)
    return lcl_p, dewpoint(vapor_pressure(lcl_p, w))


def lcl_steffensen(pressure, temperature, dewpt, max_iters=50, eps=1e-5):
    """Per-column (scalar) spelling of the same iteration: SciPy's
    _fixed_point_helper with del2 acceleration, stop rule |p - p_prev|/|p_prev| < eps
    evaluated for THIS column only.  This is the build's LCL specification (the
    reference's result depends on what else is in the dask block).
    Returns (p_lcl, t_lcl, n_iter)."""
    p_start = float(pressure)
    t = float(temperature)
    w = float(mixing_ratio(saturation_vapor_pressure(dewpt), p_start))
    p0 = p_start
    n = 0
    converged = False
    for n in range(1, max_iters + 1):
        p1 = float(_lcl_iter(p0, p_start, w, t))
        p2 = float(_lcl_iter(p1, p_start, w, t))
        d = p2 - 2.0 * p1 + p0
        p = p0 - (p1 - p0) ** 2 / d if d != 0 else p2
        relerr = (p - p0) / p0 if p0 != 0 else p
        if abs(relerr) < eps:
            converged = True
            break
        p0 = p
    if not converged:
        p = np.nan
    if np.isclose(p, p_start):
        p = p_start
    return p, float(dewpoint(vapor_pressure(p, w))), n


# -- moist adiabat ------------------------------------------------------------------
def _moist_dt_dp(p, t):
    """dT/dp of MetPy's pseudo-adiabat (metpy.calc.moist_lapse.dt)."""
    rs = saturation_mixing_ratio(p, t)
    frac = (RD * t + LV * rs) / (CP_D + (LV * LV * rs * EPSILON / (RD * t ** 2)))
    return frac / p


def moist_lapse_ode(pressure, temperature, reference_pressure=None,
                    method='LSODA', atol=1e-7, rtol=1.5e-8):
    """metpy.calc.moist_lapse (1.4.1): integrate dT/dp from the reference pressure
    to every requested pressure (both directions).  This is what the reference's
    KAT harness patches in (unit_tests.py:114-140) and what builds the lookup
    table (pf.py:480).  `temperature` scalar; returns array like `pressure`."""
    pressure = np.atleast_1d(np.asarray(pressure, dtype=np.float64))
    t0 = float(temperature)
    if reference_pressure is None:
        reference_pressure = pressure[0]
    ref = float(reference_pressure)
    out = np.full(pressure.shape, np.nan)
    if np.isnan(ref) or np.isnan(t0):
        return out
    valid = ~np.isnan(pressure)
    close = valid & np.isclose(pressure, ref)
    out[close] = t0
    above = valid & (pressure < ref) & ~close
    below = valid & (pressure > ref) & ~close
    for side, descending in ((above, True), (below, False)):
        if not side.any():
            continue
        idx = np.nonzero(side)[0]
        ps = pressure[idx]
        order = np.argsort(-ps) if descending else np.argsort(ps)
        ps_sorted = ps[order]
        res = _si.solve_ivp(lambda p, t: _moist_dt_dp(p, t), (ref, ps_sorted[-1]), [t0],
                            method=method, atol=atol, rtol=rtol, t_eval=ps_sorted)
        if not res.success:
            raise ValueError('ODE integration failed: ' + res.message)
        tmp = np.empty(len(ps))
        tmp[order] = res.y[0]
        out[idx] = tmp
    return out


# RK4 specification of the build's "exact" moist mode ---------------------------
RK4_H_MAX = 0.1   # max step in ln(p); shared by oracle/c and the HIP kernel


def _moist_dt_dlnp(x, t):
    """dT/dln p, algebraically the same ODE as _moist_dt_dp, written with one
    division (the form the C oracle and the HIP kernel use)."""
    p = np.exp(x)
    e = saturation_vapor_pressure(t)
    pe = p - e
    num = RD * t * pe + LV * EPSILON * e
    den = CP_D * RD * t * t * pe + LV * LV * EPSILON * EPSILON * e
    return RD * t * t * num / den


def rk4_substeps(dx):
    return max(1, int(np.ceil(abs(dx) / RK4_H_MAX - 1e-12)))


def moist_lapse_rk4(pressure, temperature, reference_pressure=None):
    """Build specification of the exact moist adiabat: classical RK4 in ln p,
    marching from the reference point through the requested pressures in order
    of distance on each side; each leg is cut into ceil(|dlnp|/RK4_H_MAX) equal
    substeps.  Differs from the true ODE solution by < 1e-5 K on the
    meteorological domain (tests/test_oracle_thermo.py) -- i.e. it is closer to the
    ODE than MetPy's own LSODA tolerance."""
    pressure = np.atleast_1d(np.asarray(pressure, dtype=np.float64))
    t0 = float(temperature)
    if reference_pressure is None:
        reference_pressure = pressure[0]
    ref = float(reference_pressure)
    out = np.full(pressure.shape, np.nan)
    if np.isnan(ref) or np.isnan(t0):
        return out
    valid = ~np.isnan(pressure)
    for side in (valid & (pressure <= ref), valid & (pressure > ref)):
        idx = np.nonzero(side)[0]
        if idx.size == 0:
            continue
        order = idx[np.argsort(np.abs(np.log(pressure[idx]) - np.log(ref)), kind='stable')]
        x = np.log(ref)
        t = t0
        for k in order:
            x1 = np.log(pressure[k])
            if x1 != x:
                n = rk4_substeps(x1 - x)
                h = (x1 - x) / n
                xs = x
                for _ in range(n):
                    k1 = _moist_dt_dlnp(xs, t)
                    k2 = _moist_dt_dlnp(xs + 0.5 * h, t + 0.5 * h * k1)
                    k3 = _moist_dt_dlnp(xs + 0.5 * h, t + 0.5 * h * k2)
                    k4 = _moist_dt_dlnp(xs + h, t + h * k3)
                    t = t + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
                    xs = xs + h
                x = x1
            out[k] = t
    return out
