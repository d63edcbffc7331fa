"""
xarray-facing mirror of the hot path of traupach/xarray_parcel's modules/parcel_functions.py ("pf.py").

Same function names, argument names, defaults, return structure, attrs and assert messages as the
reference; the bodies hand the arrays to libxparcel (HIP, MI355X) through numpy_api.  Only the path named by
BASELINE.json is here: the drivers, the column algorithms they use, and the table loader.  DataArrays may be
xarray's (when installed) or the small stand-in of _xr.py.

Arrays are moved to (vert_dim, ...) order, flattened to columns and sent to the GPU; dask-backed inputs are
loaded by `.values` (the vertical must be one chunk in the reference too, pf.py:564).
"""
import numpy as np

from . import numpy_api as _api
from ._lib import XParcelError
from ._xr import DataArray, Dataset, merge

VERT = 'model_level_number'


# -- DataArray plumbing ---------------------------------------------------------------------------
def _split(x, vert_dim):
    """-> (values with vert_dim first, horizontal dims, horizontal coords, vertical coordinate)."""
    if not isinstance(x, DataArray):
        return np.asarray(x, dtype=np.float64), (), {}, None
    if vert_dim in x.dims:
        other = tuple(d for d in x.dims if d != vert_dim)
        xt = x.transpose(vert_dim, *other)
        vc = np.asarray(x.coords[vert_dim]) if vert_dim in x.coords else np.arange(x.shape[x.dims.index(vert_dim)])
    else:
        other, xt, vc = tuple(x.dims), x, None
    coords = {k: np.asarray(v) for k, v in x.coords.items() if k in other}
    return np.asarray(xt.values), other, coords, vc


def _check_index(vc, msg):
    if vc is not None and len(vc) > 1:
        assert np.all(np.abs(np.diff(vc)) == 1), msg


def _horiz(values, dims, coords, attrs=None, name=None):
    return DataArray(np.asarray(values), dims=dims, coords={k: v for k, v in coords.items() if k in dims},
                     attrs=attrs or {}, name=name)


def _vert(values, vert_dim, vcoord, dims, coords, attrs=None, name=None):
    c = {k: v for k, v in coords.items() if k in dims}
    c[vert_dim] = vcoord
    return DataArray(np.asarray(values), dims=(vert_dim,) + tuple(dims), coords=c, attrs=attrs or {}, name=name)


def _np(x):
    return x.cpu().numpy() if hasattr(x, 'cpu') else np.asarray(x)


_ATTRS = {
    'cape': {'long_name': 'Convective available potential energy', 'units': 'J kg$^{-1}$'},       # pf.py:1366-1368
    'cin': {'long_name': 'Convective inhibition', 'units': 'J kg$^{-1}$'},                          # pf.py:1383-1385
    'lcl_pressure': {'long_name': 'Lifting condensation level pressure', 'units': 'hPa'},            # pf.py:669-677
    'lcl_temperature': {'long_name': 'Lifting condensation level temperature', 'units': 'K'},
    'lcl_virtual_temperature': {'long_name': 'Lifting condensation level virtual temperature', 'units': 'K'},
    'el_pressure': {'long_name': 'Equilibrium level pressure', 'units': 'hPa'},                      # pf.py:1188-1196
    'el_temperature': {'long_name': 'Equilibrium level temperature', 'units': 'K'},
    'lfc_pressure': {'long_name': 'Level of free convection pressure', 'units': 'hPa'},
    'lfc_temperature': {'long_name': 'Level of free convection temperature', 'units': 'K'},
    'pressure': {'long_name': 'Pressure at LCL'},                                                    # pf.py:889-890 (sic)
    'temperature': {'long_name': 'Temperature at LCL', 'units': 'K'},
    'virtual_temperature': {'long_name': 'Virtual temperature', 'units': 'K'},
    'environment_temperature': {'long_name': 'Environment temperature', 'units': 'K'},               # pf.py:849-852
    'environment_dewpoint': {'long_name': 'Environment dewpoint', 'units': 'K'},
    'environment_virtual_temperature': {'long_name': 'Virtual temperature', 'units': 'K'},
}
_PROFILE_KEYS = ('pressure', 'temperature', 'virtual_temperature', 'environment_temperature',
                 'environment_virtual_temperature', 'environment_dewpoint')
_LFC_KEYS = ('lfc_pressure', 'lfc_temperature', 'el_pressure', 'el_temperature')
_LCL_KEYS = ('lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature')


def _raise_like_reference(e):
    if isinstance(e, XParcelError) and e.code == -3:
        raise AssertionError('Call load_moist_adiabat_lookups first.') from e           # pf.py:60
    raise e


# -- which moist adiabat a call uses --------------------------------------------------------------------
# The reference has ONE moist_lapse: the lookup-table one (pf.py:525-607), and every call of it starts with
# lookup_tables_loaded() (pf.py:554, 56-61).  So here: unless the caller names a mode (`moist=` keyword, an extension
# of this mirror) or has switched the module default with set_moist_lapse(), every function that lifts a parcel
# moist-adiabatically uses the tables handed over by load_moist_adiabat_lookups() and raises the reference's
# 'Call load_moist_adiabat_lookups first.' when there are none.  set_moist_lapse('exact') is the counterpart of what
# the reference's own known-answer tests do (`parcel.moist_lapse = tests.metpy_moist_lapse`,
# parcel_functions_demo.ipynb cell 33, unit_tests.py:114-140): MetPy's ODE instead of the tables.
_MOIST = {'override': None}


def set_moist_lapse(mode=None):
    """Module-wide moist-adiabat mode: None (the reference's behaviour: lookup tables), 'exact' (MetPy's ODE by RK4),
    'family' (the same ODE from the adiabat-family table) or 'table'."""
    assert mode is None or mode in _api.L.MOIST, "mode must be None, 'exact', 'family' or 'table'"
    _MOIST['override'] = mode


def _moist_mode(moist=None):
    if moist is not None:
        return moist
    if _MOIST['override'] is not None:
        return _MOIST['override']
    lookup_tables_loaded()                                                               # pf.py:554
    return 'table'


def _run(pressure, temperature, dewpoint, vert_dim, parcel, depth=None, parcel_values=None, trim=False, **kwargs):
    kwargs['moist'] = _moist_mode(kwargs.get('moist'))
    p, dims, coords, vc = _split(pressure, vert_dim)
    t, _, _, _ = _split(temperature, vert_dim)
    td, _, _, _ = _split(dewpoint, vert_dim)
    _check_index(vc, 'Vert_dim index increments must all be 1.')                        # pf.py:957
    try:
        res = _api.cape_cin_columns(p, t, td, parcel=parcel, depth=depth, parcel_values=parcel_values,
                                    want_profile=True, **kwargs)
    except XParcelError as e:
        _raise_like_reference(e)
    vtc = kwargs.get('virtual_temperature_correction', True)
    cc = Dataset({k: _horiz(_np(res[k]), dims, coords, attrs=dict(_ATTRS[k]), name=k) for k in ('cape', 'cin')})
    cc.attrs = {'correction': ('Virtual temperature correction used in CAPE/CIN calculations.' if vtc else
                               'Virtual temperature correction not used in CAPE/CIN calculations.')}  # pf.py:1453, 1472
    prof_np = {k: _np(res['profile'][k]) for k in _PROFILE_KEYS}
    nrow = prof_np['pressure'].shape[0]
    if trim:
        # the reference drops levels that are NaN in every column (pf.py:1552, 1637): trim the NaN padding
        while nrow > 1 and np.all(np.isnan(prof_np['pressure'][nrow - 1])):
            nrow -= 1
    vcoord = np.arange(nrow) + (vc[0] if vc is not None else 0)    # re-indexed vertical coordinate (pf.py:875)
    profile = Dataset()
    for k in _PROFILE_KEYS:
        profile[k] = _vert(prof_np[k][:nrow], vert_dim, vcoord, dims, coords, attrs=dict(_ATTRS[k]), name=k)
    for k in _LCL_KEYS + _LFC_KEYS:
        profile[k] = _horiz(_np(res[k]), dims, coords, attrs=dict(_ATTRS[k]), name=k)
    extra = {k: _horiz(_np(res[k]), dims, coords, name=k) for k in
             ('parcel_pressure', 'parcel_temperature', 'parcel_dewpoint', 'parcel_index', 'lfc_index', 'el_index',
              'status')}
    return cc, profile, extra


def _prefix(res, prefix):
    if prefix is not None:
        res = res.rename({'cape': prefix + '_cape', 'cin': prefix + '_cin'})              # pf.py:1510-1512
    return res


# -- drivers ------------------------------------------------------------------------------------------
def cape_cin(pressure, temperature, dewpoint, parcel_temperature, parcel_pressure, parcel_dewpoint,
             vert_dim=VERT, virtual_temperature_correction=True, lcl_interp='log', **kwargs):
    """pf.py:1394.  Returns (Dataset{cape, cin}, profile Dataset merged with LFC/EL)."""
    pv = tuple(np.asarray(getattr(x, 'values', x), dtype=np.float64)
               for x in (parcel_pressure, parcel_temperature, parcel_dewpoint))
    cc, profile, _ = _run(pressure, temperature, dewpoint, vert_dim, 'explicit', parcel_values=pv,
                          virtual_temperature_correction=virtual_temperature_correction, lcl_interp=lcl_interp,
                          **kwargs)
    return cc, profile


def surface_based_cape_cin(pressure, temperature, dewpoint, vert_dim=VERT, prefix=None, **kwargs):
    """pf.py:1477."""
    res, profile, _ = _run(pressure, temperature, dewpoint, vert_dim, 'surface', **kwargs)
    res.cape.attrs['description'] = 'CAPE for surface-based parcel.'                   # pf.py:1508-1509
    res.cin.attrs['description'] = 'CIN for surface-based parcel.'
    return _prefix(res, prefix), profile


def _named(pressure, temperature, dewpoint):
    assert getattr(pressure, 'name', None) == 'pressure', 'Pressure requires name pressure.'              # pf.py:1538
    assert getattr(temperature, 'name', None) == 'temperature', 'Temperature requires name temperature.'   # pf.py:1539
    assert getattr(dewpoint, 'name', None) == 'dewpoint', 'Dewpoint requires name dewpoint.'              # pf.py:1541


def _parcel_ds(extra, long_names):
    ds = Dataset()
    for k, ln in long_names.items():
        da = extra['parcel_' + k]
        ds[k] = DataArray(da.values, dims=da.dims, coords=da.coords, attrs={'long_name': ln}, name=k)
    return ds


def most_unstable_cape_cin(pressure, temperature, dewpoint, vert_dim=VERT, depth=300, prefix=None, **kwargs):
    """pf.py:1557.  Returns (cape/cin, profile, most-unstable parcel)."""
    _named(pressure, temperature, dewpoint)
    res, profile, extra = _run(pressure, temperature, dewpoint, vert_dim, 'most_unstable', depth=depth, trim=True,
                               **kwargs)
    desc = f'most-unstable parcel in lowest {depth} hPa.'
    res.cape.attrs['description'] = f'CAPE for {desc}'
    res.cin.attrs['description'] = f'CIN for {desc}'
    layer = _parcel_ds(extra, {'pressure': 'Pressure', 'temperature': 'Temperature', 'dewpoint': 'Dewpoint'})
    return _prefix(res, prefix), profile, layer


def mixed_layer_cape_cin(pressure, temperature, dewpoint, vert_dim=VERT, depth=100, prefix=None, **kwargs):
    """pf.py:1651.  Returns (cape/cin, profile, mixed parcel)."""
    _named(pressure, temperature, dewpoint)
    res, profile, extra = _run(pressure, temperature, dewpoint, vert_dim, 'mixed_layer', depth=depth, trim=True,
                               **kwargs)
    desc = f'fully-mixed lowest {depth} hPa parcel'
    res.cape.attrs['description'] = f'CAPE for {desc}.'
    res.cin.attrs['description'] = f'CIN for {desc}'
    mp = _parcel_ds(extra, {'pressure': 'Pressure', 'temperature': 'Mixed parcel temperature',
                            'dewpoint': 'Mixed-parcel dewpoint'})
    mp.temperature.attrs['units'] = 'K'
    return _prefix(res, prefix), profile, mp


# -- column algorithms -----------------------------------------------------------------------------------
def lcl(parcel_pressure, parcel_temperature, parcel_dewpoint):
    """pf.py:609.  Dataset with lcl_pressure, lcl_temperature, lcl_virtual_temperature."""
    p, dims, coords, _ = _split(parcel_pressure, None)
    t, _, _, _ = _split(parcel_temperature, None)
    td, _, _, _ = _split(parcel_dewpoint, None)
    p, t, td = np.broadcast_arrays(p, t, td)
    r = _api.lcl(p, t, td)
    return Dataset({k: _horiz(_np(r[k]).reshape(p.shape), dims, coords, attrs=dict(_ATTRS[k]), name=k)
                    for k in _LCL_KEYS})


def dry_lapse(pressure, parcel_temperature, parcel_pressure=None, vert_dim=VERT):
    """pf.py:291."""
    p, dims, coords, vc = _split(pressure, vert_dim)
    pt = np.asarray(getattr(parcel_temperature, 'values', parcel_temperature), dtype=np.float64)
    pp = None if parcel_pressure is None else np.asarray(getattr(parcel_pressure, 'values', parcel_pressure))
    out = _np(_api.dry_lapse(p, pt, pp))
    return _vert(out, vert_dim, vc, dims, coords, attrs={'long_name': 'Dry lapse rate temperature', 'units': 'K'})


def moist_lapse(pressure, parcel_temperature, parcel_pressure=None, vert_dim=VERT, persist=True, moist=None):
    """pf.py:525: the reference's table lookup (needs load_moist_adiabat_lookups() first, pf.py:554); `moist='exact'`
    / `'family'` or set_moist_lapse() select the ODE instead (an extension of this mirror)."""
    moist = _moist_mode(moist)
    p, dims, coords, vc = _split(pressure, vert_dim)
    pt = np.asarray(getattr(parcel_temperature, 'values', parcel_temperature), dtype=np.float64)
    pp = None if parcel_pressure is None else np.asarray(getattr(parcel_pressure, 'values', parcel_pressure))
    try:
        out = _np(_api.moist_lapse(p, pt, pp, moist=moist))
    except XParcelError as e:
        _raise_like_reference(e)
    return _vert(out, vert_dim, vc, dims, coords, attrs={'long_name': 'Moist lapse rate temperature', 'units': 'K'})


def parcel_profile(pressure, parcel_pressure, parcel_temperature, parcel_dewpoint, vert_dim=VERT, moist=None):
    """pf.py:712."""
    moist = _moist_mode(moist)
    p, dims, coords, vc = _split(pressure, vert_dim)
    pv = [np.asarray(getattr(x, 'values', x), dtype=np.float64) for x in
          (parcel_pressure, parcel_temperature, parcel_dewpoint)]
    try:
        r = _api.parcel_profile(p, *pv, moist=moist)
    except XParcelError as e:
        _raise_like_reference(e)
    out = Dataset()
    out['pressure'] = _vert(p, vert_dim, vc, dims, coords, name='pressure')
    out['temperature'] = _vert(_np(r['temperature']), vert_dim, vc, dims, coords,
                               attrs={'long_name': 'Lifted parcel temperature', 'units': 'K'}, name='temperature')
    out['virtual_temperature'] = _vert(_np(r['virtual_temperature']), vert_dim, vc, dims, coords,
                                       attrs=dict(_ATTRS['virtual_temperature']), name='virtual_temperature')
    for k in _LCL_KEYS:
        out[k] = _horiz(_np(r[k]), dims, coords, attrs=dict(_ATTRS[k]), name=k)
    return out


def parcel_profile_with_lcl(pressure, temperature, dewpoint, parcel_pressure, parcel_temperature, parcel_dewpoint,
                            vert_dim=VERT, lcl_interp='log', moist=None):
    """pf.py:806."""
    pv = tuple(np.asarray(getattr(x, 'values', x), dtype=np.float64)
               for x in (parcel_pressure, parcel_temperature, parcel_dewpoint))
    _, profile, _ = _run(pressure, temperature, dewpoint, vert_dim, 'explicit', parcel_values=pv,
                         lcl_interp=lcl_interp, moist=moist)
    return Dataset({k: profile[k] for k in _PROFILE_KEYS + _LCL_KEYS})


def lfc_el(pressure, parcel_temperature, temperature, lcl_pressure, lcl_temperature, vert_dim=VERT):
    """pf.py:1066."""
    p, dims, coords, vc = _split(pressure, vert_dim)
    par, _, _, _ = _split(parcel_temperature, vert_dim)
    env, _, _, _ = _split(temperature, vert_dim)
    _check_index(vc, 'Index increments must all be 1.')                                    # pf.py:1012
    r = _api.lfc_el(p, par, env, np.asarray(getattr(lcl_pressure, 'values', lcl_pressure)),
                    np.asarray(getattr(lcl_temperature, 'values', lcl_temperature)))
    return Dataset({k: _horiz(_np(r[k]), dims, coords, attrs=dict(_ATTRS[k]), name=k) for k in _LFC_KEYS})


def cape_cin_base(pressure, temperature, lfc_pressure, el_pressure, parcel_temperature, vert_dim=VERT,
                  pos_cape_neg_cin=True, post_zero_cin=False, **kwargs):
    """pf.py:1291."""
    p, dims, coords, vc = _split(pressure, vert_dim)
    env, _, _, _ = _split(temperature, vert_dim)
    par, _, _, _ = _split(parcel_temperature, vert_dim)
    _check_index(vc, 'Index increments must all be 1.')                                    # pf.py:1221
    r = _api.cape_cin_base(p, env, np.asarray(getattr(lfc_pressure, 'values', lfc_pressure)),
                           np.asarray(getattr(el_pressure, 'values', el_pressure)), par,
                           pos_cape_neg_cin=pos_cape_neg_cin, post_zero_cin=post_zero_cin)
    res = Dataset({k: _horiz(_np(r[k]), dims, coords, attrs=dict(_ATTRS[k]), name=k) for k in ('cape', 'cin')})
    res.attrs = []                                                                          # pf.py:1391
    return res


def most_unstable_parcel(dat, depth=300, vert_dim=VERT):
    """pf.py:102: dat = Dataset with pressure, temperature, dewpoint."""
    p, dims, coords, _ = _split(dat['pressure'], vert_dim)
    t, _, _, _ = _split(dat['temperature'], vert_dim)
    td, _, _, _ = _split(dat['dewpoint'], vert_dim)
    r = _api.most_unstable_parcel(p, t, td, depth=depth)
    return Dataset({k: _horiz(_np(r[k]), dims, coords, attrs=dict(getattr(dat[k], 'attrs', {})), name=k)
                    for k in ('pressure', 'temperature', 'dewpoint')})


def mixed_parcel(pressure, temperature, dewpoint, depth=100, vert_dim=VERT):
    """pf.py:229."""
    assert getattr(pressure, 'name', 'pressure') is not None, 'pressure requires name pressure.'   # pf.py:263
    p, dims, coords, _ = _split(pressure, vert_dim)
    t, _, _, _ = _split(temperature, vert_dim)
    td, _, _, _ = _split(dewpoint, vert_dim)
    r = _api.mixed_parcel(p, t, td, depth=depth)
    mp = Dataset({k: _horiz(_np(r[k]), dims, coords, name=k) for k in ('pressure', 'temperature', 'dewpoint')})
    mp.temperature.attrs.update({'long_name': 'Mixed parcel temperature', 'units': 'K'})
    mp.dewpoint.attrs.update({'long_name': 'Mixed-parcel dewpoint'})
    return mp


def mixed_layer(dat, depth=100, vert_dim=VERT):
    """pf.py:137: dat = Dataset with pressure and the variables to mix."""
    p, dims, coords, _ = _split(dat['pressure'], vert_dim)
    arrs = {'pressure': p}
    for k in dat.keys():
        if k != 'pressure':
            arrs[k] = _split(dat[k], vert_dim)[0]
    r = _api.mixed_layer(arrs, depth=depth)
    return Dataset({k: _horiz(_np(v), dims, coords, name=k) for k, v in r.items()})


# -- SURVEY 8(f) items on the same kernels ------------------------------------------------------------------------
def wet_bulb_temperature(pressure, temperature, dewpoint, vert_dim=VERT, moist=None):
    """pf.py:389 (Normand's rule; its descent is a moist_lapse call, pf.py:436)."""
    moist = _moist_mode(moist)
    p, dims, coords, vc = _split(pressure, vert_dim)
    t, _, _, _ = _split(temperature, vert_dim)
    td, _, _, _ = _split(dewpoint, vert_dim)
    try:
        out = _np(_api.wet_bulb_temperature(p, t, td, moist=moist))
    except XParcelError as e:
        _raise_like_reference(e)
    if vc is None:
        return _horiz(out.reshape(p.shape), dims, coords, name='wet_bulb_temperature',
                      attrs={'long_name': 'Wet bulb temperature', 'units': 'K'})
    return _vert(out, vert_dim, vc, dims, coords, name='wet_bulb_temperature',
                 attrs={'long_name': 'Wet bulb temperature', 'units': 'K'})


def _interp(x, coords, at, dim, log, keep_attrs=True):
    cv, dims, hcoords, _ = _split(coords, dim)
    atv = np.asarray(getattr(at, 'values', at))

    def one(v):
        xv = _split(v, dim)[0]
        return _horiz(_np(_api.interp_level(cv, xv, atv, log=log)), dims, hcoords,
                      attrs=dict(getattr(v, 'attrs', {})) if keep_attrs else {}, name=getattr(v, 'name', None))
    if isinstance(x, Dataset):                                      # every variable of the dataset (pf.py:82, 896, 901)
        return Dataset({k: one(x[k]) for k in (list(x.data_vars) if hasattr(x, 'data_vars') else list(x.keys()))})
    return one(x)


def log_interp(x, coords, at, dim=VERT):
    """pf.py:1813: `x` a DataArray or a Dataset."""
    return _interp(x, coords, at, dim, log=True)


def linear_interp(x, coords, at, dim=VERT, keep_attrs=True, extrapolate=False):
    """pf.py:1758 (extrapolate=False only): `x` a DataArray or a Dataset."""
    assert not extrapolate, 'extrapolation is not part of the MI355X path'
    return _interp(x, coords, at, dim, log=False, keep_attrs=keep_attrs)


def lifted_index(profile, vert_dim=VERT, description=None, prefix=None):
    """pf.py:1722."""
    p, dims, coords, _ = _split(profile['pressure'], vert_dim)
    prof = {'pressure': p, 'temperature': _split(profile['temperature'], vert_dim)[0],
            'environment_temperature': _split(profile['environment_temperature'], vert_dim)[0]}
    attrs = {'long_name': 'Lifted index', 'units': 'K'}
    if description is not None:
        attrs['description'] = description
    name = 'lifted_index' if prefix is None else prefix + '_lifted_index'
    return Dataset({name: _horiz(_np(_api.lifted_index(prof)), dims, coords, attrs=attrs, name=name)})


def mixing_ratio(temperature, dewpoint, pressure):
    """pf.py:684."""
    if not isinstance(temperature, DataArray):
        return _np(_api.mixing_ratio(temperature, dewpoint, pressure))
    out = _np(_api.mixing_ratio(np.asarray(temperature.values), np.asarray(getattr(dewpoint, 'values', dewpoint)),
                                np.asarray(getattr(pressure, 'values', pressure))))
    return DataArray(out, dims=temperature.dims, coords=temperature.coords, attrs={'units': 'kg kg$^{-1}$'})


def virtual_temperature(temperature, mixing_ratio, epsilon=0.608):
    """pf.py:782."""
    res = temperature * (1 + epsilon * mixing_ratio)
    if isinstance(res, DataArray):
        res.attrs['units'] = 'K'
        res.attrs['long_name'] = 'Virtual temperature'
    return res


def wet_bulb_temperature_fast(temperature, dewpoint):
    """pf.py:364: "1/3 rule" estimate (array arithmetic on the DataArrays, as in the reference)."""
    wb = temperature - (1 / 3) * (temperature - dewpoint)
    wb.name = 'wet_bulb_temperature'
    wb.attrs['long_name'] = 'Wet bulb temperature'
    wb.attrs['description'] = 'Estimated using 1/3 method.'
    wb.attrs['units'] = 'K'
    return wb


def deep_convective_index(pressure, temperature, dewpoint, lifted_index, vert_dim=VERT, description=None, prefix=None):
    """pf.py:1830 (Kunz 2009): T + Td at 850 hPa [deg C] minus the lifted index."""
    p, dims, coords, _ = _split(pressure, vert_dim)
    li = _split(lifted_index, vert_dim)[0]
    dci = _np(_api.deep_convective_index(p, _split(temperature, vert_dim)[0], _split(dewpoint, vert_dim)[0], li))
    attrs = {'long_name': 'Deep convective index', 'units': 'C'}
    if description is not None:
        attrs['description'] = description
    name = 'dci' if prefix is None else prefix + '_dci'
    return Dataset({name: _horiz(dci, dims, coords, attrs=attrs, name=name)})


def lapse_rate(pressure, temperature, height, from_pressure=700, to_pressure=500, vert_dim=VERT):
    """pf.py:2102: observed lapse rate between two pressure levels [K/km]."""
    p, dims, coords, _ = _split(pressure, vert_dim)
    out = _np(_api.lapse_rate(p, _split(temperature, vert_dim)[0], _split(height, vert_dim)[0],
                              from_pressure=from_pressure, to_pressure=to_pressure))
    return _horiz(out, dims, coords, attrs={'long_name': 'Lapse rate',
                                            'description': f'{from_pressure}-{to_pressure} hPa lapse rate',
                                            'units': 'K km$^{-1}$'})


def freezing_level_height(temperature, height, vert_dim=VERT):
    """pf.py:2137: height of the lowest 273.15 K crossing of the temperature profile."""
    t, dims, coords, vc = _split(temperature, vert_dim)
    _check_index(vc, 'Index increments must all be 1.')                                 # pf.py:1011
    out = _np(_api.freezing_level_height(t, _split(height, vert_dim)[0]))
    return _horiz(out, dims, coords, name='freezing_level',
                  attrs={'long_name': 'Freezing-level height',
                         'description': 'Height of zero degree dry-bulb temperature isotherm.', 'units': 'm'})


def melting_level_height(pressure, temperature, dewpoint, height, fast=True, vert_dim=VERT, moist=None):
    """pf.py:2160: freezing level of the wet-bulb temperature; returns (melting level, wet bulb)."""
    if fast:
        wb = wet_bulb_temperature_fast(temperature=temperature, dewpoint=dewpoint)
    else:
        wb = wet_bulb_temperature(pressure=pressure, temperature=temperature, dewpoint=dewpoint, vert_dim=vert_dim,
                                  moist=moist)
    mlh = freezing_level_height(temperature=wb, height=height, vert_dim=vert_dim)
    mlh.attrs['long_name'] = 'Melting-level height'
    mlh.attrs['description'] = 'Height of zero degree wet-bulb temperature isotherm.'
    mlh.name = 'melting_level'
    return mlh, wb


def isobar_temperature(pressure, temperature, isobar, vert_dim=VERT):
    """pf.py:2193."""
    p, dims, coords, _ = _split(pressure, vert_dim)
    out = _np(_api.isobar_temperature(p, _split(temperature, vert_dim)[0], isobar))
    return _horiz(out, dims, coords, attrs={'description': f'Temperature at {isobar} hPa.',
                                            'long_name': 'Isobar temperature', 'units': 'K'})


def dewpoint_from_specific_humidity(pressure, temperature, specific_humidity, vert_dim=VERT):
    """metpy.calc.dewpoint_from_specific_humidity (MetPy 1.4.1 chain) as the reference's harness and products call it
    (parcel_test.py:262-266, pf.py:1889-1894), result in K."""
    p, dims, coords, vc = _split(pressure, vert_dim)
    out = _np(_api.dewpoint_from_specific_humidity(p, _split(temperature, vert_dim)[0],
                                                   _split(specific_humidity, vert_dim)[0]))
    attrs = {'long_name': 'Dewpoint temperature', 'units': 'K'}
    if vc is None:
        return _horiz(out.reshape(p.shape), dims, coords, name='dewpoint', attrs=attrs)
    return _vert(out, vert_dim, vc, dims, coords, name='dewpoint', attrs=attrs)


# -- product bundle (pf.py:1951-2100, 2216-2407) ------------------------------------------------------------------------
def wind_shear(surface_wind_u, surface_wind_v, wind_u, wind_v, height, shear_height=6000, vert_dim=VERT):
    """pf.py:2216: Dataset with shear_u, shear_v, shear_magnitude [m/s] and positive_shear."""
    h, dims, coords, _ = _split(height, vert_dim)
    r = _api.wind_shear(_split(surface_wind_u, vert_dim)[0], _split(surface_wind_v, vert_dim)[0],
                        _split(wind_u, vert_dim)[0], _split(wind_v, vert_dim)[0], h, shear_height=shear_height)
    names = {'shear_u': f'Surface to {shear_height} m wind shear, U component.',
             'shear_v': f'Surface to {shear_height} m wind shear, V component.',
             'shear_magnitude': f'Surface to {shear_height} m bulk wind shear.',
             'positive_shear': f'True if {shear_height} wind > surface wind.'}
    out = Dataset()
    for k, ln in names.items():
        attrs = {'long_name': ln}
        if k != 'positive_shear':
            attrs['units'] = 'm s$^{-1}$'
        out[k] = _horiz(_np(r[k]), dims, coords, attrs=attrs, name=k)
    return out


def significant_hail_parameter(mucape, mixing_ratio, lapse, temp_500, shear, flh):
    """pf.py:2261 (SHIP)."""
    vals = [np.asarray(getattr(x, 'values', x), dtype=np.float64) for x in (mucape, mixing_ratio, lapse, temp_500, shear, flh)]
    ship = _api.significant_hail_parameter(*vals)
    ref = mucape if isinstance(mucape, DataArray) else None
    return DataArray(ship, dims=ref.dims if ref is not None else None, coords=ref.coords if ref is not None else None,
                     attrs={'long_name': 'Significant hail parameter', 'units': 'J kg$^{-2}$ g K$^2$ km$^{-1}$ m s$^{-1}$'})


def valid_data(dat, vert_dim):
    """pf.py:2308."""
    vc = np.asarray(dat[vert_dim].values if hasattr(dat[vert_dim], 'values') else dat[vert_dim])
    assert np.all(np.abs(np.diff(vc)) == 1), 'Index increments must all be 1.'
    p, _, _, _ = _split(dat['pressure'], vert_dim)
    assert np.nanmax(np.diff(p, axis=0)) < 0, 'Pressures must decrease with increasing level number.'
    return True


_BUNDLE_ATTRS = {
    'mu_mixing_ratio': {'long_name': 'Mixing ratio', 'description': 'Mixing ratio of most unstable parcel'},
    'lapse_rate_700_500': {'long_name': 'Lapse rate', 'description': '700-500 hPa lapse rate', 'units': 'K km$^{-1}$'},
    'temp_500': {'description': 'Temperature at 500 hPa.', 'long_name': 'Isobar temperature', 'units': 'K'},
    'freezing_level': {'long_name': 'Freezing-level height', 'units': 'm',
                       'description': 'Height of zero degree dry-bulb temperature isotherm.'},
    'melting_level': {'long_name': 'Melting-level height', 'units': 'm',
                      'description': 'Height of zero degree wet-bulb temperature isotherm.'},
}


def conv_properties(dat, vert_dim=VERT, ignore_nans=False, moist=None):
    """pf.py:1951: the convective-property bundle.  `dat` holds pressure, temperature, specific_humidity, height_asl on
    `vert_dim`, wind_u, wind_v, wind_height_above_surface on their own vertical, surface_wind_u, surface_wind_v."""
    return _bundle(dat, vert_dim, _api.conv_properties, ignore_nans=ignore_nans, moist=_moist_mode(moist))


def _bundle(dat, vert_dim, fn, **kw):
    p, dims, coords, _ = _split(dat['pressure'], vert_dim)
    wdim = [d for d in dat['wind_u'].dims if d not in dims][0]
    arrs = {k: _split(dat[k], vert_dim)[0] for k in ('pressure', 'temperature', 'specific_humidity', 'height_asl')}
    arrs.update({k: _split(dat[k], wdim)[0] for k in ('wind_u', 'wind_v', 'wind_height_above_surface')})
    arrs.update({k: _split(dat[k], vert_dim)[0] for k in ('surface_wind_u', 'surface_wind_v')})
    try:
        r = fn(arrs, **kw)
    except XParcelError as e:
        _raise_like_reference(e)
    out = Dataset()
    for k, v in r.items():
        attrs = dict(_BUNDLE_ATTRS.get(k, {}))
        for base in ('cape', 'cin'):
            if k.endswith('_' + base):
                attrs = dict(_ATTRS[base])
        if k.endswith('_lifted_index'):
            attrs = {'long_name': 'Lifted index', 'units': 'K'}
        if k.endswith('_dci'):
            attrs = {'long_name': 'Deep convective index', 'units': 'C'}
        out[k] = _horiz(_np(v), dims, coords, attrs=attrs, name=k)
    return out


def min_conv_properties(dat, vert_dim=VERT, moist=None):
    """pf.py:1873: the minimal property set (mixed-layer CAPE/CIN + lifted index, lapse rate, T500, freezing / melting
    level, 0-6 km shear)."""
    return _bundle(dat, vert_dim, _api.min_conv_properties, moist=_moist_mode(moist))


def storm_proxies(dat):
    """pf.py:2323: proxies (booleans) and SHIP from the Dataset returned by conv_properties()."""
    ref = dat['mu_cape']
    r = _api.storm_proxies({k: np.asarray(dat[k].values) for k in dat.keys()})
    labels = {'proxy_Craven2004': 'Craven 2004', 'proxy_Kunz2007': 'Kunz 2007', 'proxy_Trapp2007': 'Trapp 2007',
              'proxy_Marsh2009': 'Marsh 2009', 'proxy_Allen2011': 'Allen 2011', 'proxy_Allen2014': 'Allen 2014',
              'proxy_Eccel2012': 'Eccel 2012', 'proxy_Mohr2013': 'Mohr 2013', 'proxy_SHIP_0.1': 'SHIP > 0.1'}
    out = Dataset()
    for k, v in r.items():
        attrs = {'long_name': 'Proxy ' + labels[k]} if k in labels else {'long_name': 'Significant hail parameter (SHIP)',
                                                                         'units': 'J kg$^{-2}$ g K$^2$ km$^{-1}$ m s$^{-1}$'}
        out[k] = DataArray(np.asarray(v), dims=ref.dims, coords=ref.coords, attrs=attrs, name=k)
    return out



# -- the reference's array primitives ------------------------------------------------------------------------------------
# (the CAPE / CIN kernels do not use them -- they stream a column once -- but callers of the reference can)
def _vars(ds):
    return list(ds.data_vars) if hasattr(ds, 'data_vars') else list(ds.keys())


def _ds_split(ds, vert_dim):
    """Dataset -> (dict name -> values with vert_dim first, horizontal dims, their coords, vertical coordinate)."""
    arrs, dims, coords, vc = {}, (), {}, None
    for k in _vars(ds):
        v, d, c, vcoord = _split(ds[k], vert_dim)
        arrs[k] = v
        if vc is None and vcoord is not None:
            dims, coords, vc = d, c, vcoord
    shape = next((v.shape for v in arrs.values() if vc is not None and v.ndim == len(dims) + 1), None)
    for k, v in arrs.items():                              # variables without the vertical broadcast along it (xarray.broadcast)
        if shape is not None and v.shape != shape:
            arrs[k] = np.broadcast_to(v, shape)
    return arrs, dims, coords, vc


def _ds_vert(arrs, vert_dim, vcoord, dims, coords, like=None):
    return Dataset({k: _vert(_np(v), vert_dim, vcoord, dims, coords, name=k,
                             attrs=dict(getattr(like[k], 'attrs', {})) if like is not None and k in like else {})
                    for k, v in arrs.items()})


def _per_point(x, dims):
    """DataArray / scalar on the horizontal dims -> plain array."""
    if isinstance(x, DataArray):
        other = tuple(d for d in x.dims if d in dims)
        return np.asarray(x.transpose(*other).values if other else x.values)
    return np.asarray(x)


def round_to(x, to, dp=2):
    """pf.py:358."""
    return np.round(np.round(x / to) * to, dp)


def interp1d_numba(at, xp, fp, out=None):
    """pf.py:23: numpy.interp along the LAST axis of each argument (the reference's gufunc signature
    (m),(n),(n)->(m)), leading axes broadcast; `out` is the gufunc's optional output array."""
    at, xp, fp = (np.asarray(getattr(v, 'values', v), dtype=np.float64) for v in (at, xp, fp))
    lead = np.broadcast_shapes(at.shape[:-1], xp.shape[:-1], fp.shape[:-1])
    m, n = at.shape[-1], xp.shape[-1]
    a = np.moveaxis(np.broadcast_to(at, lead + (m,)), -1, 0).reshape(m, -1)
    shared = xp.ndim == 1
    x = xp if shared else np.moveaxis(np.broadcast_to(xp, lead + (n,)), -1, 0).reshape(n, -1)
    f = fp if fp.ndim == 1 else np.moveaxis(np.broadcast_to(fp, lead + (n,)), -1, 0).reshape(n, -1)
    res = np.moveaxis(_np(_api.interp1d(a, x, f)).reshape((m,) + lead), 0, -1)
    if out is not None:
        out[...] = res
        return out
    return res


def bound_pressure(pressure, bound, vert_dim=VERT):
    """pf.py:208."""
    p, dims, coords, _ = _split(pressure, vert_dim)
    return _horiz(_np(_api.bound_pressure(p, _per_point(bound, dims))), dims, coords, attrs=dict(getattr(pressure, 'attrs', {})),
                  name=getattr(pressure, 'name', None))


def get_layer(dat, depth=100, vert_dim=VERT, interpolate=True):
    """pf.py:63."""
    arrs, dims, coords, vc = _ds_split(dat, vert_dim)
    r = _api.get_layer(arrs, depth=depth, interpolate=interpolate)
    n = len(vc)
    vcoord = (np.arange(n + 1) + vc[0]) if interpolate else vc              # insert_level re-indexes (pf.py:971)
    return _ds_vert(r, vert_dim, vcoord, dims, coords, like=dat)


def insert_level(d, level, coords, vert_dim=VERT, fill_value=-999):
    """pf.py:933."""
    arrs, dims, hcoords, vc = _ds_split(d, vert_dim)
    _check_index(vc, 'Vert_dim index increments must all be 1.')                           # pf.py:957
    assert not np.any(arrs[coords] == fill_value), 'dataset d contains fill_value.'        # pf.py:965
    lev = {k: np.broadcast_to(_per_point(level[k], dims), arrs[coords].shape[1:]) for k in _vars(level)}
    r = _api.insert_level(arrs, lev, coords=coords, fill_value=fill_value)
    return _ds_vert(r, vert_dim, np.arange(len(vc) + 1) + vc[0], dims, hcoords, like=d)


def find_intersections(x, a, b, dim, log_x=False):
    """pf.py:992."""
    xv, dims, coords, vc = _split(x, dim)
    _check_index(vc, 'Index increments must all be 1.')                                    # pf.py:1012
    av, bv = _split(a, dim)[0], _split(b, dim)[0]
    r = _api.find_intersections(xv, np.broadcast_to(av, xv.shape), np.broadcast_to(bv, xv.shape), log_x=log_x)
    return _ds_vert(r, 'offset_dim', vc[1:], dims, coords)                                 # pf.py:1062


def trapz(dat, x, dim, mask=None, only_positive=False, only_negative=False):
    """pf.py:164: `dat` a Dataset, `x` the NAME of its x variable; every variable is integrated (x itself included, as
    in the reference)."""
    arrs, dims, coords, vc = _ds_split(dat, dim)
    _check_index(vc, 'Index increments must all be 1.')                                    # pf.py:183
    assert not (only_positive and only_negative), 'Only negative OR positive regions can be included in trapz.'
    m = None
    if mask is not None:
        m = _split(mask, dim)[0] if isinstance(mask, DataArray) else np.asarray(mask)
        m = np.broadcast_to(m, (m.shape[0],) + arrs[x].shape[1:])[:len(vc) - 1]            # labels 0 .. n-2 (pf.py:190-195)
    r = _api.trapz(arrs, arrs[x], mask=m, only_positive=only_positive, only_negative=only_negative)
    return Dataset({k: _horiz(_np(v), dims, coords, name=k) for k, v in r.items()})


def trap_around_zeros(x, y, dim, log_x=True, start=0):
    """pf.py:1200 (start = 0)."""
    xv, dims, coords, vc = _split(x, dim)
    _check_index(vc, 'Index increments must all be 1.')                                    # pf.py:1221
    areas, mask = _api.trap_around_zeros(xv, _split(y, dim)[0], log_x=log_x, start=start)
    labels = np.concatenate([vc, vc[1:]])                                                  # pf.py:1273: concat of the two families
    return (_ds_vert(areas, dim, labels, dims, coords),
            _vert(_np(mask), dim, vc, dims, coords))


def shift_out_nans(x, name, dim):
    """pf.py:1699."""
    arrs, dims, coords, vc = _ds_split(x, dim)
    _check_index(vc, 'Index increments must all be 1.')                                    # pf.py:1712
    return _ds_vert(_api.shift_out_nans(arrs, name), dim, vc, dims, coords, like=x)


def from_most_unstable_parcel(pressure, temperature, dewpoint, vert_dim=VERT, depth=300):
    """pf.py:1517."""
    _named(pressure, temperature, dewpoint)
    p, dims, coords, vc = _split(pressure, vert_dim)
    rp, rt, rtd, parcel, kept = _api.from_most_unstable_parcel(p, _split(temperature, vert_dim)[0], _split(dewpoint, vert_dim)[0],
                                                               depth=depth)
    vcoord = vc[kept]                                                                      # dropna keeps the labels (pf.py:1552)
    outs = [_vert(_np(v), vert_dim, vcoord, dims, coords, attrs=dict(getattr(src, 'attrs', {})), name=k)
            for k, v, src in (('pressure', rp, pressure), ('temperature', rt, temperature), ('dewpoint', rtd, dewpoint))]
    layer = Dataset({k: _horiz(_np(parcel[k]), dims, coords, name=k) for k in ('pressure', 'temperature', 'dewpoint')})
    return outs[0], outs[1], outs[2], layer


def mix_layer(pressure, temperature, dewpoint, vert_dim=VERT, depth=100, load=True):
    """pf.py:1604."""
    _named(pressure, temperature, dewpoint)
    p, dims, coords, vc = _split(pressure, vert_dim)
    rp, rt, rtd, parcel, kept = _api.mix_layer(p, _split(temperature, vert_dim)[0], _split(dewpoint, vert_dim)[0], depth=depth)
    surv = vc[kept]
    vcoord = np.concatenate([[(surv.min() if len(surv) else vc[0]) - 1], surv])            # pf.py:1641
    outs = [_vert(_np(v), vert_dim, vcoord, dims, coords, attrs=dict(getattr(src, 'attrs', {})), name=k)
            for k, v, src in (('pressure', rp, pressure), ('temperature', rt, temperature), ('dewpoint', rtd, dewpoint))]
    mp = Dataset({k: _horiz(_np(parcel[k]), dims, coords, name=k) for k in ('pressure', 'temperature', 'dewpoint')})
    mp.temperature.attrs.update({'long_name': 'Mixed parcel temperature', 'units': 'K'})
    mp.dewpoint.attrs.update({'long_name': 'Mixed-parcel dewpoint'})
    return outs[0], outs[1], outs[2], mp


def add_lcl_to_profile(profile, vert_dim=VERT, environment=None, interpolator='log'):
    """pf.py:858."""
    assert interpolator in ['linear', 'log'], 'interpolator must be linear or log'         # pf.py:878
    lev_keys = ('pressure', 'temperature', 'virtual_temperature')
    p, dims, coords, vc = _split(profile['pressure'], vert_dim)
    _check_index(vc, 'Vert_dim index increments must all be 1.')
    prof = {k: _split(profile[k], vert_dim)[0] for k in lev_keys}
    for k in _LCL_KEYS:
        prof[k] = _per_point(profile[k], dims)
    env = None
    if environment is not None:
        env = _ds_split(environment, vert_dim)[0]
    r = _api.add_lcl_to_profile(prof, environment=env, interpolator=interpolator)
    vcoord = np.arange(len(vc) + 1) + vc[0]
    out = Dataset()
    for k, v in r.items():
        if k in _LCL_KEYS:
            out[k] = _horiz(_np(v), dims, coords, attrs=dict(_ATTRS[k]), name=k)
        elif k.startswith('environment_'):
            out[k] = _vert(_np(v), vert_dim, vcoord, dims, coords, attrs=dict(getattr(environment[k[12:]], 'attrs', {})), name=k)
        else:
            out[k] = _vert(_np(v), vert_dim, vcoord, dims, coords, attrs=dict(getattr(profile[k], 'attrs', {})), name=k)
    out['temperature'].attrs['long_name'] = 'Temperature at LCL'                           # pf.py:889-891 (sic)
    out['pressure'].attrs['long_name'] = 'Pressure at LCL'
    out['lcl_virtual_temperature'].attrs['long name'] = 'Virtual temperature at LCL'
    return out


def moist_adiabat_lookup(pressure_levels=np.round(np.arange(1100, 2, step=-0.5), 1),
                         temperatures=np.round(np.arange(173, 316, step=0.02), 2), pres_step=0.5, temp_step=0.02):
    """pf.py:447: the two lookup tables as Datasets in the reference's layout -- adiabat_lookup.adiabat(pressure,
    temperature) = adiabat number (NaN = none) and adiabats.temperature(adiabat, pressure) -- generated on the GPU
    (adiabat_tables.moist_adiabat_lookup).  Only the reference's default grid is implemented: the device tables are
    addressed arithmetically on it."""
    from . import adiabat_tables as at
    pl, tt = at._grids()
    assert (np.array_equal(np.asarray(pressure_levels), pl) and np.array_equal(np.asarray(temperatures), tt) and
            pres_step == at.P_STEP and temp_step == at.T_STEP), 'only the default table grid is implemented'
    return _table_datasets(*at.moist_adiabat_lookup())


def _table_datasets(index, adiabats):
    from . import adiabat_tables as at
    pl, tt = at._grids()
    lookup = Dataset({'adiabat': DataArray(np.where(index == 0, np.nan, index.astype(np.float64)), dims=('pressure', 'temperature'),
                                           coords={'pressure': pl, 'temperature': tt}, attrs={'long_name': 'Adiabat index'},
                                           name='adiabat')})
    curves = Dataset({'temperature': DataArray(np.asarray(adiabats, dtype=np.float64)[:, ::-1], dims=('adiabat', 'pressure'),
                                               coords={'adiabat': np.arange(1, adiabats.shape[0] + 1), 'pressure': pl},
                                               attrs={'long_name': 'Temperature', 'units': 'K'}, name='temperature')})
    return lookup, curves


def moist_adiabat_tables(regenerate=False, cache=True, chunks=None, base_dir='.',
                         lookup_cache='/adiabat_lookups/moist_adiabat_lookup.nc',
                         adiabats_cache='/adiabat_lookups/adiabats_cache.nc', **kwargs):
    """pf.py:318: the cached tables, or freshly generated ones.  The cache is one .npz under
    base_dir/adiabat_lookups/ (NetCDF is not available here; `lookup_cache` / `adiabats_cache` / `chunks` are accepted
    and ignored); unlike the reference, a missing cache is regenerated instead of failing to open."""
    from . import adiabat_tables as at
    return _table_datasets(*at.moist_adiabat_tables(regenerate=regenerate, cache=cache, base_dir=base_dir))


# -- tables (pf.py:39-61) ------------------------------------------------------------------------------
def load_moist_adiabat_lookups(**kwargs):
    """pf.py:39: load (or generate and cache) the reference-format lookup tables; from here on every moist call of this
    module uses them, as in the reference."""
    from . import adiabat_tables
    adiabat_tables.load_moist_adiabat_lookups(**kwargs)


def lookup_tables_loaded():
    """pf.py:56."""
    from . import _lib
    assert _lib.load().xp_tables_loaded(), 'Call load_moist_adiabat_lookups first.'
