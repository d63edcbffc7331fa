"""
oracle/parcel_oracle.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

One-column-at-a-time NumPy restatement of the reference's parcel-lifting path
(traupach/xarray_parcel modules/parcel_functions.py = "pf.py").  Every function
here takes plain 1-D float64 arrays along the vertical (level 0 = surface) and
mirrors the *array* semantics of the reference function it cites -- including
its where/shift/insert gymnastics, skip-NaN reductions and label alignment --
so that it can serve as the checker for the HIP kernels, which are organised
completely differently (one streaming pass per column, LCL as a virtual level).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path never does.

The moist adiabat is selectable (set_moist_lapse / the `moist` argument):
  'ode'   MetPy's LSODA solve, what the reference's KAT harness patches in
          (unit_tests.py:114-140); default for the KAT tests
  'rk4'   the build's exact-mode specification (oracle/thermo.py moist_lapse_rk4)
  'table' the reference's lookup tables (pf.py:525-607), see oracle/tables.py
"""
import warnings

import numpy as np

from . import thermo as th

_MOIST = {'mode': 'ode', 'tables': None}


def set_moist_lapse(mode, tables=None):
    """Select the moist-adiabat implementation used by moist_lapse()."""
    assert mode in ('ode', 'rk4', 'table', 'family')
    _MOIST['mode'] = mode
    if tables is not None:
        _MOIST['tables'] = tables


# -- skip-NaN reductions with xarray semantics (sum of nothing = 0, max of nothing = NaN)
def _nmax(x):
    x = np.asarray(x, dtype=np.float64)
    x = x[~np.isnan(x)]
    return x.max() if x.size else np.nan


def _nmin(x):
    x = np.asarray(x, dtype=np.float64)
    x = x[~np.isnan(x)]
    return x.min() if x.size else np.nan


def _nsum(x):
    x = np.asarray(x, dtype=np.float64)
    return float(np.nansum(x))


def _nmean(x):
    x = np.asarray(x, dtype=np.float64)
    x = x[~np.isnan(x)]
    return x.mean() if x.size else np.nan


def _where(cond, x, other=np.nan):
    return np.where(cond, x, other)


def _f(x):
    return np.asarray(x, dtype=np.float64)


# -- L1 array primitives ----------------------------------------------------------
def insert_level(d, level, coords='pressure', fill_value=-999.0):
    """pf.py:933-990.  `d`: dict name -> 1-D array (all same length n); `level`:
    dict name -> scalar (keys of `level` define the output keys).  The new level
    goes after every level whose coordinate is >= the new coordinate; an existing
    equal coordinate therefore stays *below* it (pf.py:950-954).  NaN coordinates
    are treated as 'above' through the fill value (pf.py:962-966)."""
    n = len(d[coords])
    c = _f(d[coords])
    assert not np.any(c == fill_value), 'dataset d contains fill_value.'
    nan_rows = np.isnan(c)
    dd = {k: np.where(nan_rows, fill_value, _f(v)) for k, v in d.items()}
    lev_c = level[coords]
    with np.errstate(invalid='ignore'):
        below_m = dd[coords] >= lev_c
        above_m = dd[coords] < lev_c
    out = {}
    for k in level.keys():
        below = np.full(n + 1, np.nan)
        below[:n] = np.where(below_m, dd[k], np.nan)
        above = np.full(n + 1, np.nan)
        above[1:] = np.where(above_m, dd[k], np.nan)
        out[k] = (below, above)
    coord_below = out[coords][0]
    res = {}
    for k in level.keys():
        below, above = out[k]
        merged = np.where(np.isnan(coord_below), above, below)   # pf.py:977
        res[k] = merged
    coord_merged = res[coords].copy()
    for k in level.keys():
        res[k] = np.where(np.isnan(coord_merged), level[k], res[k])  # pf.py:985
        res[k] = np.where(res[k] == fill_value, np.nan, res[k])      # pf.py:988
    return res


def linear_interp(x, coords, at):
    """pf.py:1758-1811 (extrapolate=False).  `x`: dict of arrays or an array."""
    coords = _f(coords)
    with np.errstate(invalid='ignore'):
        cb = _nmin(_where(coords >= at, coords))
        ca = _nmax(_where(coords <= at, coords))

    def one(v):
        v = _f(v)
        xb = _nmean(_where(coords == cb, v))
        xa = _nmean(_where(coords == ca, v))
        with np.errstate(invalid='ignore', divide='ignore'):
            res = xb + (xa - xb) * ((at - cb) / (ca - cb))
        return xb if xb == xa else res

    if isinstance(x, dict):
        return {k: one(v) for k, v in x.items()}
    return one(x)


def log_interp(x, coords, at):
    """pf.py:1813-1828."""
    with np.errstate(invalid='ignore', divide='ignore'):
        return linear_interp(x, np.log(_f(coords)), np.log(at))


def find_intersections(x, a, b, log_x=False):
    """pf.py:992-1064.  Returns a dict of arrays of length n-1; entry i describes
    the interval between levels i and i+1 (reference label i+1 on 'offset_dim')."""
    x = _f(x)
    a = _f(a)
    b = _f(b)
    with np.errstate(invalid='ignore', divide='ignore'):
        if log_x:
            x = np.log(x)
        diffs = np.diff(np.sign(a - b))                       # pf.py:1019
        after = np.where(diffs == 0, 0.0, 1.0)                # NaN diffs -> 1 (pf.py:1022)
        flagged = after == 1
        sign_change = np.sign(_where(flagged, a[1:]) - _where(flagged, b[1:]))
        x0 = _where(flagged, x[:-1])
        x1 = _where(flagged, x[1:])
        a0 = _where(flagged, a[:-1])
        a1 = _where(flagged, a[1:])
        b0 = _where(flagged, b[:-1])
        b1 = _where(flagged, b[1:])
        dy0 = a0 - b0
        dy1 = a1 - b1
        ix = (dy1 * x0 - dy0 * x1) / (dy1 - dy0)              # pf.py:1046
        iy = ((ix - x0) / (x1 - x0)) * (a1 - a0) + a0         # pf.py:1050
        if log_x:
            ix = np.exp(ix)
        inc = sign_change > 0
        dec = sign_change < 0
    return {'all_intersect_x': ix, 'all_intersect_y': iy,
            'increasing_x': _where(inc, ix), 'increasing_y': _where(inc, iy),
            'decreasing_x': _where(dec, ix), 'decreasing_y': _where(dec, iy)}


def lfc_el(pressure, parcel_temperature, temperature, lcl_pressure, lcl_temperature):
    """pf.py:1066-1198.  Returns dict(lfc_pressure, lfc_temperature, el_pressure,
    el_temperature) plus the bookkeeping the build also exports: lfc_index /
    el_index = index i of the interval (levels i, i+1 of the profile handed in)
    holding the chosen crossing, -1 if none, -2 if the LFC was replaced by the LCL."""
    p = _f(pressure)
    par = _f(parcel_temperature)
    env = _f(temperature)
    n = len(p)
    inter = find_intersections(p, par, env, log_x=True)
    above_raw = find_intersections(p[1:], par[1:], env[1:], log_x=True)
    # reindex_like(intersections): interval 0 has no entry in the 'above' set.
    inter_above = {k: np.concatenate([[np.nan], v]) for k, v in above_raw.items()}
    if not (env[0] != par[0]):                                # pf.py:1117-1120
        inter = inter_above
    with np.errstate(invalid='ignore'):
        above_lcl = inter['increasing_x'] < lcl_pressure
        lfc_p = _nmax(_where(above_lcl, inter['increasing_x']))
        lfc_t = _nmax(_where(inter['increasing_x'] == lfc_p, inter['increasing_y']))
        el_p = _nmin(inter_above['decreasing_x'])
        el_t = _nmax(_where(inter['decreasing_x'] == el_p, inter_above['decreasing_y']))

        temps_available = ~np.isnan(par) & ~np.isnan(env)
        top_pressure = p == _nmin(_where(temps_available, p))
        top_prof_temp = _nmax(_where(top_pressure, par))
        top_env_temp = _nmax(_where(top_pressure, env))
        status_top_nan = bool(np.isnan(top_env_temp) != np.isnan(_nmax(env)))  # assert pf.py:1149
        top_colder = top_prof_temp <= top_env_temp
        el_above_lcl = el_p < lcl_pressure
        el_exists = bool(top_colder and el_above_lcl)
        if not el_exists:
            el_p = np.nan
            el_t = np.nan

        lfc_missing = np.isnan(_nmax(inter['increasing_x']))
        above = p < lcl_pressure
        pos_parcel = bool(np.any(_where(above, par) > _where(above, env)))
        no_lfc_pos_parcel = pos_parcel and lfc_missing
        exists_but_na = (not lfc_missing) and np.isnan(lfc_p)
        lfc_below_el_above = bool(exists_but_na and (el_p < lcl_pressure))
        replace = no_lfc_pos_parcel or lfc_below_el_above
        lfc_idx = -1
        if not np.isnan(lfc_p):
            lfc_idx = int(np.nonzero(inter['increasing_x'] == lfc_p)[0][0])
        el_idx = -1
        if not np.isnan(el_p):
            el_idx = int(np.nonzero(inter_above['decreasing_x'] == el_p)[0][-1])
        if replace:
            lfc_p = lcl_pressure
            lfc_t = lcl_temperature
            lfc_idx = -2
    return {'lfc_pressure': float(lfc_p), 'lfc_temperature': float(lfc_t),
            'el_pressure': float(el_p), 'el_temperature': float(el_t),
            'lfc_index': lfc_idx, 'el_index': el_idx, 'status_top_nan': status_top_nan}


def trapz(dat, x, mask=None, only_positive=False, only_negative=False):
    """pf.py:164-206 for one variable.  `dat`, `x`: arrays of n levels (already
    masked to NaN outside the wanted layer); mask: n-1 booleans (interval i kept)."""
    assert not (only_positive and only_negative)
    dat = _f(dat)
    x = _f(x)
    dx = np.abs(np.diff(x))
    means = 0.5 * (dat[:-1] + dat[1:])                        # rolling(2).mean, min_periods=2
    if mask is not None:
        dx = _where(mask, dx)
        means = _where(mask, means)
    areas = dx * means
    with np.errstate(invalid='ignore'):
        if only_positive:
            areas = _where(areas > 0, areas)
        if only_negative:
            areas = _where(areas < 0, areas)
    return _nsum(areas)


def trap_around_zeros(x, y, log_x=True):
    """pf.py:1200-1289 with start=0.  Returns (areas dict over 2n-1 entries, mask of
    n-1 intervals that do NOT hold a valid zero crossing)."""
    x = _f(x)
    y = _f(y)
    n = len(x)
    zi = find_intersections(x, y, np.zeros(n), log_x=log_x)
    zero_y = zi['all_intersect_y']                            # length n-1, interval i
    zero_x = zi['all_intersect_x']
    with np.errstate(invalid='ignore', divide='ignore'):
        if log_x:
            x = np.log(x)
            zero_x = np.log(zero_x)
    valid = ~np.isnan(zero_y)                                 # after_zeros_mask, labels 1..n-1
    # 'before' family: defined on levels 0..n-1; level k is just before a zero in (k,k+1)
    before_mask = np.concatenate([valid, [False]])
    a_area = np.full(n, np.nan)
    a_x = np.full(n, np.nan)
    a_dx = np.full(n, np.nan)
    for k in range(n - 1):
        if before_mask[k]:
            dx = x[k] - zero_x[k]
            a_area[k] = (y[k] / 2.0) * abs(dx)
            a_x[k] = x[k] - dx / 2.0
            a_dx[k] = abs(dx)
    # 'after' family: defined on labels 1..n-1 (level k+1 just after a zero in (k,k+1))
    b_area = np.full(n - 1, np.nan)
    b_x = np.full(n - 1, np.nan)
    b_dx = np.full(n - 1, np.nan)
    for k in range(n - 1):
        if valid[k]:
            dx = x[k + 1] - zero_x[k]
            b_area[k] = (y[k + 1] / 2.0) * abs(dx)
            b_x[k] = x[k + 1] - dx / 2.0
            b_dx[k] = abs(dx)
    areas = {'area': np.concatenate([a_area, b_area]),
             'x': np.concatenate([a_x, b_x]),
             'dx': np.concatenate([a_dx, b_dx])}
    mask = np.isnan(a_area)[:n - 1]                           # pf.py:1285-1287
    return areas, mask


def cape_cin_base(pressure, temperature, lfc_pressure, el_pressure, parcel_temperature,
                  pos_cape_neg_cin=True, post_zero_cin=False):
    """pf.py:1291-1392."""
    p = _f(pressure)
    env = _f(temperature)
    par = _f(parcel_temperature)
    if np.isnan(el_pressure):
        el_pressure = _nmin(p)                                # pf.py:1329
    temp_diff = par - env
    with np.errstate(invalid='ignore', divide='ignore'):
        logp = np.log(p)
        areas, trapz_mask = trap_around_zeros(p, temp_diff, log_x=True)
        ax = np.exp(areas['x'])

        in_layer = (p <= lfc_pressure) & (p >= el_pressure)
        sel = (ax <= lfc_pressure) & (ax >= el_pressure)
        a = _where(sel, areas['area'])
        if pos_cape_neg_cin:
            a = _where(a > 0, a)
        cape = th.RD * trapz(_where(in_layer, temp_diff), _where(in_layer, logp),
                             mask=trapz_mask, only_positive=pos_cape_neg_cin)
        cape = cape + th.RD * _nsum(a)

        below = p >= lfc_pressure
        sel = ax >= lfc_pressure
        a = _where(sel, areas['area'])
        if pos_cape_neg_cin:
            a = _where(a < 0, a)
        cin = th.RD * trapz(_where(below, temp_diff), _where(below, logp),
                            mask=trapz_mask, only_negative=pos_cape_neg_cin)
        cin = cin + th.RD * _nsum(a)
    if post_zero_cin and not (cin <= 0):
        cin = 0.0
    return {'cape': float(cape), 'cin': float(cin)}


# -- L2 column algorithms ---------------------------------------------------------------
def dry_lapse(pressure, parcel_temperature, parcel_pressure=None):
    """pf.py:291-316."""
    p = _f(pressure)
    if parcel_pressure is None:
        parcel_pressure = _nmax(p)
    return parcel_temperature * (p / parcel_pressure) ** th.KAPPA


def moist_lapse(pressure, parcel_temperature, parcel_pressure=None, moist=None):
    """pf.py:525-607 (table mode) / unit_tests.py:114-140 (ODE mode, the KAT harness)."""
    mode = moist or _MOIST['mode']
    p = _f(np.atleast_1d(pressure))
    if parcel_pressure is None:
        parcel_pressure = p[0]
    if mode == 'ode':
        return th.moist_lapse_ode(p, parcel_temperature, parcel_pressure)
    if mode == 'rk4':
        return th.moist_lapse_rk4(p, parcel_temperature, parcel_pressure)
    if mode == 'family':
        from . import family
        return family.moist_lapse_family(p, parcel_temperature, parcel_pressure)
    from . import tables
    return tables.moist_lapse_table(_MOIST['tables'], p, parcel_temperature, parcel_pressure)


def mixing_ratio(temperature, dewpt, pressure):
    """pf.py:684-710."""
    rh = th.relative_humidity_from_dewpoint(temperature, dewpt)
    return th.mixing_ratio_from_relative_humidity(pressure, temperature, rh)


def virtual_temperature(temperature, mixing, epsilon=th.VT_EPSILON):
    """pf.py:782-804."""
    return temperature * (1 + epsilon * mixing)


def lcl(parcel_pressure, parcel_temperature, parcel_dewpoint, per_column=False):
    """pf.py:609-682.  per_column=False: MetPy/SciPy fixed point on the scalar (what a
    1-column KAT sees); True: the build's per-column Steffensen spelling."""
    if np.isnan(parcel_pressure) or np.isnan(parcel_temperature) or np.isnan(parcel_dewpoint):
        return {'lcl_pressure': np.nan, 'lcl_temperature': np.nan,
                'lcl_virtual_temperature': np.nan}
    if per_column:
        pl, tl, _ = th.lcl_steffensen(parcel_pressure, parcel_temperature, parcel_dewpoint)
    else:
        pl, tl = th.lcl_metpy(parcel_pressure, parcel_temperature, parcel_dewpoint)
        pl = float(pl)
        tl = float(tl)
    w = mixing_ratio(tl, tl, pl)
    return {'lcl_pressure': pl, 'lcl_temperature': tl,
            'lcl_virtual_temperature': float(virtual_temperature(tl, w))}


def parcel_profile(pressure, parcel_pressure, parcel_temperature, parcel_dewpoint,
                   moist=None, per_column_lcl=False):
    """pf.py:712-780."""
    p = _f(pressure)
    out = {'pressure': p}
    out.update(lcl(parcel_pressure, parcel_temperature, parcel_dewpoint, per_column_lcl))
    below_lcl = dry_lapse(p, parcel_temperature, parcel_pressure)
    parcel_w = mixing_ratio(parcel_temperature, parcel_dewpoint, parcel_pressure)
    above_lcl = moist_lapse(p, out['lcl_temperature'], out['lcl_pressure'], moist=moist)
    with np.errstate(invalid='ignore'):
        w = th.saturation_mixing_ratio(p, above_lcl)
        out['temperature'] = np.where(p >= out['lcl_pressure'], below_lcl, above_lcl)
        w = np.where(p <= out['lcl_pressure'], w, parcel_w)
    out['virtual_temperature'] = virtual_temperature(out['temperature'], w)
    return out


def add_lcl_to_profile(profile, environment=None, interpolator='log'):
    """pf.py:858-931."""
    assert interpolator in ('linear', 'log'), 'interpolator must be linear or log'
    level = {'pressure': profile['lcl_pressure'], 'temperature': profile['lcl_temperature'],
             'virtual_temperature': profile['lcl_virtual_temperature']}
    d = {k: profile[k] for k in ('pressure', 'temperature', 'virtual_temperature')}
    out = insert_level(d, level)
    for k in ('lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature'):
        out[k] = profile[k]
    if environment is not None:
        envd = {k: _f(v) for k, v in environment.items() if k != 'pressure'}
        interp = linear_interp if interpolator == 'linear' else log_interp
        il = interp(envd, environment['pressure'], level['pressure'])
        il['pressure'] = level['pressure']
        if 'virtual_temperature' in il:
            w = mixing_ratio(il['temperature'], il['dewpoint'], il['pressure'])
            il['virtual_temperature'] = virtual_temperature(il['temperature'], w)
        env_full = dict(envd)
        env_full['pressure'] = _f(environment['pressure'])
        new_env = insert_level(env_full, il)
        for k in envd:
            out['environment_' + k] = new_env[k]
    return out


def parcel_profile_with_lcl(pressure, temperature, dewpt, parcel_pressure, parcel_temperature,
                            parcel_dewpoint, lcl_interp='log', moist=None,
                            per_column_lcl=False):
    """pf.py:806-856."""
    profile = parcel_profile(pressure, parcel_pressure, parcel_temperature, parcel_dewpoint,
                             moist=moist, per_column_lcl=per_column_lcl)
    t = _f(temperature)
    td = _f(dewpt)
    w = mixing_ratio(t, td, profile['pressure'])
    environment = {'temperature': t, 'virtual_temperature': virtual_temperature(t, w),
                   'dewpoint': td, 'pressure': profile['pressure']}
    return add_lcl_to_profile(profile, environment=environment, interpolator=lcl_interp)


# -- L3 drivers ----------------------------------------------------------------------------
def cape_cin(pressure, temperature, dewpt, parcel_temperature, parcel_pressure, parcel_dewpoint,
             virtual_temperature_correction=True, lcl_interp='log', moist=None,
             per_column_lcl=False, **kwargs):
    """pf.py:1394-1475.  Returns (dict cape/cin, profile dict merged with lfc/el)."""
    profile = parcel_profile_with_lcl(pressure, temperature, dewpt, parcel_pressure,
                                      parcel_temperature, parcel_dewpoint,
                                      lcl_interp=lcl_interp, moist=moist,
                                      per_column_lcl=per_column_lcl)
    if not virtual_temperature_correction:
        par, env, lt = (profile['temperature'], profile['environment_temperature'],
                        profile['lcl_temperature'])
    else:
        par, env, lt = (profile['virtual_temperature'],
                        profile['environment_virtual_temperature'],
                        profile['lcl_virtual_temperature'])
    le = lfc_el(profile['pressure'], par, env, profile['lcl_pressure'], lt)
    cc = cape_cin_base(profile['pressure'], env, le['lfc_pressure'], le['el_pressure'], par,
                       **kwargs)
    profile.update(le)
    return cc, profile


def surface_based_cape_cin(pressure, temperature, dewpt, **kwargs):
    """pf.py:1477-1514."""
    p = _f(pressure)
    t = _f(temperature)
    td = _f(dewpt)
    return cape_cin(p, t, td, parcel_temperature=t[0], parcel_pressure=p[0],
                    parcel_dewpoint=td[0], **kwargs)


def bound_pressure(pressure, bound):
    """pf.py:208-227."""
    p = _f(pressure)
    diffs = np.abs(p - bound)
    return _nmax(_where(diffs == _nmin(diffs), p))


def get_layer(dat, depth=100, interpolate=True):
    """pf.py:63-100.  `dat`: dict with 'pressure' + variables."""
    p = _f(dat['pressure'])
    bottom = _nmax(p)
    if interpolate:
        top = bottom - depth
        il = log_interp({k: v for k, v in dat.items()}, p, top)
        il['pressure'] = top
        dat = insert_level(dat, il)
    else:
        top = bound_pressure(p, bottom - depth)
    pp = _f(dat['pressure'])
    with np.errstate(invalid='ignore'):
        keep = (pp <= bottom) & (pp >= top)
    return {k: _where(keep, _f(v)) for k, v in dat.items()}


def most_unstable_parcel(pressure, temperature, dewpt, depth=300):
    """pf.py:102-135.  Returns dict(pressure, temperature, dewpoint, index)."""
    layer = get_layer({'pressure': _f(pressure), 'temperature': _f(temperature),
                       'dewpoint': _f(dewpt)}, depth=depth, interpolate=False)
    with np.errstate(invalid='ignore', divide='ignore'):
        eq = th.equivalent_potential_temperature(layer['pressure'], layer['temperature'],
                                                 layer['dewpoint'])
        max_eq = _nmax(eq)
        pres = _nmax(_where(eq == max_eq, layer['pressure']))
        sel = layer['pressure'] == pres
    idx = int(np.nonzero(sel)[0][0]) if np.any(sel) else -1
    return {'pressure': _nmax(_where(sel, layer['pressure'])),
            'temperature': _nmax(_where(sel, layer['temperature'])),
            'dewpoint': _nmax(_where(sel, layer['dewpoint'])), 'index': idx}


def _drop_and_shift(dat):
    """dropna(dim, how='all') then shift_out_nans (pf.py:1552-1553, 1637-1638,
    1699-1720) for ONE column: on a grid dropna only removes levels that are NaN in
    every column; per column, what reaches cape_cin is the surviving levels followed
    by NaN padding, which is inert."""
    keys = list(dat.keys())
    allnan = np.all(np.stack([np.isnan(_f(dat[k])) for k in keys]), axis=0)
    out = {k: _f(dat[k])[~allnan] for k in keys}
    while len(out['pressure']) and np.isnan(out['pressure'][0]):
        out = {k: np.concatenate([v[1:], [np.nan]]) for k, v in out.items()}
        if np.all(np.isnan(out['pressure'])):
            break
    return out


def most_unstable_cape_cin(pressure, temperature, dewpt, depth=300, **kwargs):
    """pf.py:1517-1602."""
    p = _f(pressure)
    t = _f(temperature)
    td = _f(dewpt)
    mu = most_unstable_parcel(p, t, td, depth=depth)
    with np.errstate(invalid='ignore'):
        keep = p <= mu['pressure']
    dat = _drop_and_shift({'pressure': _where(keep, p), 'temperature': _where(keep, t),
                           'dewpoint': _where(keep, td)})
    cc, profile = cape_cin(dat['pressure'], dat['temperature'], dat['dewpoint'],
                           parcel_temperature=mu['temperature'], parcel_pressure=mu['pressure'],
                           parcel_dewpoint=mu['dewpoint'], **kwargs)
    return cc, profile, mu


def mixed_layer(dat, depth=100):
    """pf.py:137-162: layer means by trapezoid in linear p."""
    layer = get_layer(dat, depth=depth)
    lp = layer['pressure']
    pressure_depth = abs(_nmin(lp) - _nmax(lp))
    return {k: (1.0 / pressure_depth) * trapz(v, lp) for k, v in layer.items() if k != 'pressure'}


def mixed_parcel(pressure, temperature, dewpt, depth=100):
    """pf.py:229-289."""
    p = _f(pressure)
    t = _f(temperature)
    td = _f(dewpt)
    p_start = p[0]
    theta = th.potential_temperature(p, t)
    w = th.saturation_mixing_ratio(p, td)
    mp = mixed_layer({'pressure': p, 'theta': theta, 'mixing_ratio': w}, depth=depth)
    mp['temperature'] = mp['theta'] * th.exner_function(p_start)
    mp['vapour_pressure'] = th.vapor_pressure(p_start, mp['mixing_ratio'])
    mp['dewpoint'] = float(th.dewpoint(mp['vapour_pressure']))
    mp['pressure'] = p_start
    return mp


def mix_layer(pressure, temperature, dewpt, depth=100):
    """pf.py:1604-1649."""
    p = _f(pressure)
    t = _f(temperature)
    td = _f(dewpt)
    mp = mixed_parcel(p, t, td, depth=depth)
    with np.errstate(invalid='ignore'):
        keep = p < (_nmax(p) - depth)
    dat = _drop_and_shift({'pressure': _where(keep, p), 'temperature': _where(keep, t),
                           'dewpoint': _where(keep, td)})
    return (np.concatenate([[mp['pressure']], dat['pressure']]),
            np.concatenate([[mp['temperature']], dat['temperature']]),
            np.concatenate([[mp['dewpoint']], dat['dewpoint']]), mp)


def mixed_layer_cape_cin(pressure, temperature, dewpt, depth=100, **kwargs):
    """pf.py:1651-1697."""
    p, t, td, mp = mix_layer(pressure, temperature, dewpt, depth=depth)
    cc, profile = cape_cin(p, t, td, parcel_temperature=mp['temperature'],
                           parcel_pressure=mp['pressure'], parcel_dewpoint=mp['dewpoint'],
                           **kwargs)
    return cc, profile, mp


# -- "next" items that reuse the same primitives -------------------------------------
def lifted_index(profile):
    """pf.py:1722-1756 on a profile dict holding pressure, temperature, environment_temperature."""
    dat = log_interp({'environment_temperature': profile['environment_temperature'],
                      'temperature': profile['temperature']}, profile['pressure'], 500.0)
    return dat['environment_temperature'] - dat['temperature']


def wet_bulb_temperature(pressure, temperature, dewpt, moist=None):
    """pf.py:389-445: Normand's rule, level by level."""
    p = _f(pressure)
    t = _f(temperature)
    td = _f(dewpt)
    out = np.full(p.shape, np.nan)
    for k in range(len(p)):
        l = lcl(p[k], t[k], td[k])
        out[k] = moist_lapse(np.array([p[k]]), l['lcl_temperature'], l['lcl_pressure'],
                             moist=moist)[0]
    return out


def wet_bulb_temperature_fast(temperature, dewpt):
    """pf.py:364-387 ("1/3 rule")."""
    return _f(temperature) - (1 / 3) * (_f(temperature) - _f(dewpt))


def freezing_level_height(temperature, height):
    """pf.py:2137-2158: lowest x among all intersections of the temperature profile with 273.15 K."""
    t = _f(temperature)
    ix = find_intersections(_f(height), t, np.full(t.shape, 273.15))['all_intersect_x']
    return _nmin(ix)


def melting_level_height(pressure, temperature, dewpt, height, fast=True, moist=None):
    """pf.py:2160-2191."""
    wb = wet_bulb_temperature_fast(temperature, dewpt) if fast else \
        wet_bulb_temperature(pressure, temperature, dewpt, moist=moist)
    return freezing_level_height(wb, height), wb


def isobar_temperature(pressure, temperature, isobar):
    """pf.py:2193-2214."""
    return log_interp(temperature, pressure, isobar)


def lapse_rate(pressure, temperature, height, from_pressure=700, to_pressure=500):
    """pf.py:2102-2135 [K/km]."""
    t0 = log_interp(temperature, pressure, from_pressure)
    t1 = log_interp(temperature, pressure, to_pressure)
    z0 = log_interp(height, pressure, from_pressure) / 1000
    z1 = log_interp(height, pressure, to_pressure) / 1000
    with np.errstate(invalid='ignore', divide='ignore'):
        return (t1 - t0) / (z1 - z0)


def deep_convective_index(pressure, temperature, dewpt, li):
    """pf.py:1830-1870 (Kunz 2009) [deg C]."""
    dat = log_interp({'temperature': temperature, 'dewpoint': dewpt}, pressure, 850.0)
    return (dat['temperature'] - 273.15) + (dat['dewpoint'] - 273.15) - li


def surface_cape_vector(pressure, temperature, specific_humidity, **kwargs):
    """parcel_test.py:250-274: q -> dewpoint (MetPy 1.4.1 chain), then surface-based CAPE / CIN."""
    td = th.dewpoint_from_specific_humidity(_f(pressure), _f(temperature), _f(specific_humidity))
    return surface_based_cape_cin(pressure, temperature, td, **kwargs)



# -- product bundle (pf.py:1951-2100, 2216-2407), one column at a time -------------------------------------------------
def wind_shear(surface_wind_u, surface_wind_v, wind_u, wind_v, height, shear_height=6000):
    """pf.py:2216-2259."""
    hu = linear_interp(wind_u, height, shear_height)
    hv = linear_interp(wind_v, height, shear_height)
    su, sv = hu - surface_wind_u, hv - surface_wind_v
    with np.errstate(invalid='ignore'):
        return {'shear_u': su, 'shear_v': sv, 'shear_magnitude': np.sqrt(su ** 2 + sv ** 2),
                'positive_shear': bool(np.sqrt(hu ** 2 + hv ** 2) > np.sqrt(surface_wind_u ** 2 + surface_wind_v ** 2))}


def significant_hail_parameter(mucape, mixing_ratio, lapse, temp_500, shear, flh):
    """pf.py:2261-2306 on scalars or arrays."""
    with np.errstate(invalid='ignore'):
        mixing_ratio = np.asarray(mixing_ratio, dtype=np.float64) * 1e3
        lapse = -np.asarray(lapse, dtype=np.float64)
        temp_500 = np.asarray(temp_500, dtype=np.float64) - 273.15
        shear = np.asarray(shear, dtype=np.float64)
        mucape = np.asarray(mucape, dtype=np.float64)
        flh = np.asarray(flh, dtype=np.float64)
        shear = _where(shear <= 27, _where(shear >= 7, shear))
        mixing_ratio = _where(mixing_ratio <= 13.6, _where(mixing_ratio >= 11, mixing_ratio))
        temp_500 = np.where(temp_500 <= -5.5, temp_500, -5.5)
        ship = mucape * mixing_ratio * lapse * -temp_500 * shear / 42000000
        ship = np.where(mucape >= 1300, ship, ship * (mucape / 1300))
        ship = np.where(lapse >= 5.8, ship, ship * (lapse / 5.8))
        ship = np.where(flh >= 2400, ship, ship * (flh / 2400))
    return ship


def conv_properties(pressure, temperature, specific_humidity, height_asl, surface_wind_u, surface_wind_v, wind_u, wind_v,
                    wind_height_above_surface, ignore_nans=False, moist=None):
    """pf.py:1951-2100 for ONE column; returns a dict of scalars with the reference's variable names."""
    p, t, q = _f(pressure), _f(temperature), _f(specific_humidity)
    with np.errstate(all='ignore'):
        td = th.dewpoint_from_specific_humidity(p, t, q)
    valid = not (np.isnan(td).any() or np.isnan(p).any() or np.isnan(t).any() or np.isnan(q).any())
    if not ignore_nans and not valid:                     # blanked at the end anyway (pf.py:2097-2098): skip the work
        names = [f'{a}_{b}' for a in ('mu', 'mixed_100', 'mixed_50') for b in ('cape', 'cin', 'lifted_index', 'dci')]
        names += ['mu_mixing_ratio', 'lapse_rate_700_500', 'temp_500', 'freezing_level', 'melting_level', 'shear_u', 'shear_v',
                  'shear_magnitude']
        return dict({k: np.nan for k in names}, positive_shear=False)
    out = {}
    cc, prof, parcel = most_unstable_cape_cin(p, t, td, depth=250, moist=moist)
    out['mu_cape'], out['mu_cin'] = cc['cape'], cc['cin']
    w = th.saturation_mixing_ratio(parcel['pressure'], parcel['dewpoint'])     # specific_humidity_from_dewpoint ...
    qs = w / (1.0 + w)
    out['mu_mixing_ratio'] = qs / (1.0 - qs)                                    # ... mixing_ratio_from_specific_humidity
    out['mu_lifted_index'] = lifted_index(prof)
    for depth in (100, 50):
        cc, prof, _ = mixed_layer_cape_cin(p, t, td, depth=depth, moist=moist)
        out[f'mixed_{depth}_cape'], out[f'mixed_{depth}_cin'] = cc['cape'], cc['cin']
        out[f'mixed_{depth}_lifted_index'] = lifted_index(prof)
    for pre in ('mu', 'mixed_100', 'mixed_50'):
        out[pre + '_dci'] = deep_convective_index(p, t, td, out[pre + '_lifted_index'])
    out['lapse_rate_700_500'] = lapse_rate(p, t, height_asl)
    out['temp_500'] = isobar_temperature(p, t, 500.0)
    out['freezing_level'] = freezing_level_height(t, height_asl)
    out['melting_level'] = melting_level_height(p, t, td, height_asl)[0]
    out.update(wind_shear(surface_wind_u, surface_wind_v, wind_u, wind_v, wind_height_above_surface))
    return out


def min_conv_properties(pressure, temperature, specific_humidity, height_asl, surface_wind_u, surface_wind_v, wind_u, wind_v,
                        wind_height_above_surface, moist=None):
    """pf.py:1873-1949 for ONE column."""
    p, t, q = _f(pressure), _f(temperature), _f(specific_humidity)
    with np.errstate(all='ignore'):
        td = th.dewpoint_from_specific_humidity(p, t, q)
    cc, prof, _ = mixed_layer_cape_cin(p, t, td, depth=100, moist=moist)
    out = {'mixed_100_cape': cc['cape'], 'mixed_100_cin': cc['cin'], 'mixed_100_lifted_index': lifted_index(prof),
           'lapse_rate_700_500': lapse_rate(p, t, height_asl), 'temp_500': isobar_temperature(p, t, 500.0),
           'freezing_level': freezing_level_height(t, height_asl),
           'melting_level': melting_level_height(p, t, td, height_asl)[0]}
    out.update(wind_shear(surface_wind_u, surface_wind_v, wind_u, wind_v, wind_height_above_surface))
    return out


def storm_proxies(dat):
    """pf.py:2323-2407 on a dict of arrays (the output of conv_properties for many columns)."""
    d = {k: np.asarray(v, dtype=np.float64) if k != 'positive_shear' else np.asarray(v, dtype=bool) for k, v in dat.items()}
    with np.errstate(invalid='ignore'):
        s06 = d['shear_magnitude']
        c100 = _where(d['mixed_100_cape'] >= 0, d['mixed_100_cape'])
        c50 = _where(d['mixed_50_cape'] >= 0, d['mixed_50_cape'])
        mucape = _where(d['mu_cape'] >= 0, d['mu_cape'])
        out = {}
        out['proxy_Craven2004'] = (c100 * s06) >= 20000
        out['proxy_Kunz2007'] = np.logical_or(d['mixed_100_lifted_index'] <= -2.07,
                                              np.logical_or(mucape >= 1474, d['mixed_100_dci'] >= 25.7))
        tr = np.logical_and(c100 * s06 >= 10000, c100 >= 100)
        tr = np.logical_and(tr, s06 >= 5)
        out['proxy_Trapp2007'] = np.logical_and(tr, d['positive_shear'])
        out['proxy_Marsh2009'] = (c100 * s06) >= 10000
        out['proxy_Allen2011'] = c50 * s06 ** 1.67 >= 25000
        al = np.logical_and(out['proxy_Allen2011'], d['mixed_50_cin'] > -25)
        al = np.logical_and(al, s06 > 7.5)
        out['proxy_Allen2014'] = np.logical_and(al, d['lapse_rate_700_500'] < -6.5)
        out['proxy_Eccel2012'] = np.logical_and(c100 * s06 > 10000, d['mixed_100_cin'] > -50)
        mo = np.logical_or(d['mixed_100_lifted_index'] <= -1.6, c100 >= 439)
        out['proxy_Mohr2013'] = np.logical_or(mo, d['mixed_100_dci'] >= 26.4)
        out['ship'] = significant_hail_parameter(mucape, d['mu_mixing_ratio'], d['lapse_rate_700_500'], d['temp_500'], s06,
                                                 d['freezing_level'])
        out['proxy_SHIP_0.1'] = out['ship'] > 0.1
    return out

warnings.filterwarnings('ignore', message='Mean of empty slice')
