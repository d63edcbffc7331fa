"""
Counterpart of the timing harness of traupach/xarray_parcel's modules/parcel_test.py ("pt.py") for the MI355X path:
time_function (pt.py:18-35), compare / compare_results (pt.py:37-66, 577-584), surface_cape_vector (pt.py:250-274),
conv_properties_xarray (pt.py:416-547: every function of the path on one dataset, the vectorised leg of the reference's
self-test) and benchmark_cape (pt.py:586-619).

The reference's `surface_cape_serial` leg (MetPy, one column at a time) has no counterpart here -- MetPy is not part of
this build -- so benchmark_cape reports the two legs that exist: 'xr_load' (host arrays in, host arrays out: includes
the PCIe copies, like the reference's timed region includes materialisation) and 'device' (inputs already resident
in HBM as torch tensors).
"""
import time

import numpy as np

from . import numpy_api as _api
from . import parcel_functions as parcel
from ._xr import DataArray, Dataset, merge


def time_function(func, dat, **kwargs):
    """pt.py:18-35: run func(dat) and return (result, seconds); results are host arrays already (no .load())."""
    start = time.perf_counter()
    ret = func(dat, **kwargs)
    if hasattr(ret, 'load'):
        ret = ret.load()
    end = time.perf_counter()
    return ret, end - start


def compare(x, y, name, tolerance=1e-5):
    """pt.py:37-66: compare DataArray x to the reference y; prints a line when the largest difference reaches the
    tolerance or the NaN patterns differ, returns whether it stayed below."""
    xv, yv = np.broadcast_arrays(np.asarray(x.values, dtype=np.float64), np.asarray(y.values, dtype=np.float64))
    with np.errstate(all='ignore'):
        diffs = np.abs(xv - yv)
        max_rel_diff = np.round(np.nanmax(diffs / yv * 100) if np.isfinite(diffs).any() else np.nan, 2)
        max_diff = np.round(np.nanmax(diffs) if np.isfinite(diffs).any() else np.nan, 5)
    comp = bool(max_diff < tolerance)
    if not comp:
        attrs = getattr(x, 'attrs', {})
        name_and_unit = (attrs['long_name'] + ' [' + attrs.get('units', '?') + '] (' + name + ')') if 'long_name' in attrs else name + ' [?]'
        md, mr = str(max_diff) + ' ' + attrs.get('units', ''), str(max_rel_diff) + '%'
        print(f'{name_and_unit:65} {md:20} {mr:20}')
    if not np.array_equal(np.isnan(xv), np.isnan(yv)):
        print(f'NaNs differ in {name}')
    return comp


def compare_results(set1, set2):
    """pt.py:577-584."""
    print(f'{"Differences":65} {"Max abs. diff":20} {"Max rel. diff":20}')
    for variable in set2.keys():
        compare(set1[variable], set2[variable], name=variable)


def conv_properties_xarray(dat, vert_dim='model_level_number', virt_temp=True, lcl_interp='log', pos_cape_neg_cin=False,
                           post_zero_cin=False):
    """pt.py:416-547: the convective properties the reference's self-test compares with MetPy, all through the mirror --
    dewpoint from specific humidity, the 100 hPa mixed parcel, dry and moist adiabats from the surface, mixed-layer /
    most-unstable / surface-based CAPE and CIN, the surface parcel's profile with LFC / EL, lifted index, deep
    convective index, wet-bulb temperature (Normand's rule and the 1/3 rule).  Returns one Dataset with the
    reference's variable names."""
    dat['dewpoint'] = parcel.dewpoint_from_specific_humidity(pressure=dat['pressure'], temperature=dat['temperature'],
                                                             specific_humidity=dat['specific_humidity'], vert_dim=vert_dim)
    mp = parcel.mixed_parcel(pressure=dat['pressure'], temperature=dat['temperature'], dewpoint=dat['dewpoint'], vert_dim=vert_dim)
    mp = mp.rename({'pressure': 'mp_pressure', 'temperature': 'mp_temperature', 'dewpoint': 'mp_dewpoint'})
    t0 = dat['temperature'].isel({vert_dim: 0})
    dry = parcel.dry_lapse(pressure=dat['pressure'], parcel_temperature=t0, vert_dim=vert_dim)
    dry.name = 'dry_lapse_temp'
    moist = parcel.moist_lapse(pressure=dat['pressure'], parcel_temperature=t0, parcel_pressure=900, vert_dim=vert_dim)
    moist.name = 'moist_lapse_temp'
    opts = dict(virtual_temperature_correction=virt_temp, lcl_interp=lcl_interp, pos_cape_neg_cin=pos_cape_neg_cin,
                post_zero_cin=post_zero_cin, vert_dim=vert_dim)
    mixed_cape_cin, mixed_profile, _ = parcel.mixed_layer_cape_cin(pressure=dat['pressure'], temperature=dat['temperature'],
                                                                   dewpoint=dat['dewpoint'], depth=100, prefix='mixed', **opts)
    max_cape_cin, _, _ = parcel.most_unstable_cape_cin(pressure=dat['pressure'], temperature=dat['temperature'],
                                                       dewpoint=dat['dewpoint'], depth=300, prefix='max', **opts)
    surface_profile = parcel.parcel_profile_with_lcl(pressure=dat['pressure'], temperature=dat['temperature'],
                                                     dewpoint=dat['dewpoint'], parcel_temperature=t0,
                                                     parcel_pressure=dat['pressure'].isel({vert_dim: 0}),
                                                     parcel_dewpoint=dat['dewpoint'].isel({vert_dim: 0}), vert_dim=vert_dim,
                                                     lcl_interp=lcl_interp)
    surface_lfc_el = parcel.lfc_el(pressure=surface_profile['pressure'], parcel_temperature=surface_profile['temperature'],
                                   temperature=surface_profile['environment_temperature'],
                                   lcl_pressure=surface_profile['lcl_pressure'], lcl_temperature=surface_profile['lcl_temperature'],
                                   vert_dim=vert_dim)
    surface_lfc_el = surface_lfc_el.rename({'lfc_pressure': 'surface_lfc_pressure', 'lfc_temperature': 'surface_lfc_temp',
                                            'el_pressure': 'surface_el_pressure', 'el_temperature': 'surface_el_temp'})
    surface_cape_cin, surface_profile = parcel.surface_based_cape_cin(pressure=dat['pressure'], temperature=dat['temperature'],
                                                                      dewpoint=dat['dewpoint'], prefix='surface', **opts)
    lifted_index = parcel.lifted_index(profile=mixed_profile, vert_dim=vert_dim)
    dci = parcel.deep_convective_index(pressure=dat['pressure'], temperature=dat['temperature'], dewpoint=dat['dewpoint'],
                                       lifted_index=lifted_index['lifted_index'], vert_dim=vert_dim)
    wb = parcel.wet_bulb_temperature(pressure=dat['pressure'], temperature=dat['temperature'], dewpoint=dat['dewpoint'],
                                     vert_dim=vert_dim)
    wb.name = 'wet_bulb_temperature'
    wb_fast = parcel.wet_bulb_temperature_fast(temperature=dat['temperature'], dewpoint=dat['dewpoint'])
    wb_fast.name = 'wet_bulb_temperature_fast'
    surface_profile = surface_profile.rename({'pressure': 'surf_pres', 'temperature': 'surface_profile',
                                              'lcl_pressure': 'surface_lcl_pressure', 'lcl_temperature': 'surface_lcl_temp',
                                              'environment_temperature': 'surf_temp'})
    surface_profile = _rename_dim(surface_profile, vert_dim, vert_dim + '_lcl')
    dewpoint = dat['dewpoint']
    dewpoint.name = 'dewpoint'
    return merge([dewpoint, mp, dry, moist, mixed_cape_cin, max_cape_cin, surface_profile, surface_lfc_el, surface_cape_cin,
                  lifted_index, dci, wb, wb_fast])


def _rename_dim(ds, old, new):
    """Dataset.rename of a DIMENSION (pt.py:528) for xarray and for the stand-in of _xr.py."""
    if hasattr(ds, 'data_vars'):  # pragma: no cover - real xarray
        return ds.rename({old: new})
    out = Dataset(attrs=ds.attrs)
    for k in ds.keys():
        v = ds[k]
        out[k] = DataArray(v.values, dims=tuple(new if d == old else d for d in v.dims),
                           coords={(new if c == old else c): cv for c, cv in v.coords.items()}, attrs=v.attrs, name=v.name)
    return out


def surface_cape_vector(dat, fused=True):
    """pt.py:250-274: dat holds pressure [hPa], temperature [K], specific_humidity [kg/kg]; returns the CAPE / CIN
    Dataset of surface_based_cape_cin.  fused=True converts q -> dewpoint inside the CAPE kernel (XP_HUM_SPECIFIC);
    fused=False runs the two steps of the reference one after the other."""
    if fused:
        out, _ = parcel.surface_based_cape_cin(pressure=dat['pressure'], temperature=dat['temperature'],
                                               dewpoint=dat['specific_humidity'], humidity='specific')
    else:
        dewpoint = parcel.dewpoint_from_specific_humidity(pressure=dat['pressure'], temperature=dat['temperature'],
                                                          specific_humidity=dat['specific_humidity'])
        out, _ = parcel.surface_based_cape_cin(pressure=dat['pressure'], temperature=dat['temperature'],
                                               dewpoint=dewpoint)
    return out


def _device_cape(arrs):
    import torch
    r = _api.cape_cin_columns(arrs[0], arrs[1], arrs[2], want=('cape', 'cin'), humidity='specific')
    torch.cuda.synchronize()
    return r


def benchmark_cape(dat, points=[2, 4, 8, 16, 32, 64, 101], vert_dim=parcel.VERT):
    """pt.py:586-619: wall-clock time of surface-based CAPE / CIN over the first p x p columns of `dat` (dims
    (vert_dim, latitude, longitude)) for each p in `points`."""
    import torch
    num_points, xr_load_times, device_times = [], [], []
    for p in points:
        pts = Dataset({k: dat[k].isel(latitude=slice(0, p), longitude=slice(0, p))
                       for k in ('pressure', 'temperature', 'specific_humidity')})
        _, t_host = time_function(func=surface_cape_vector, dat=pts)
        arrs = [torch.as_tensor(np.ascontiguousarray(pts[k].transpose(vert_dim, 'latitude', 'longitude').values)).cuda()
                for k in ('pressure', 'temperature', 'specific_humidity')]
        torch.cuda.synchronize()
        _, t_dev = time_function(func=_device_cape, dat=arrs)
        num_points.append(p * p)
        xr_load_times.append(t_host)
        device_times.append(t_dev)
    return Dataset({'xr_load': DataArray(np.asarray(xr_load_times), dims=('pts',), coords={'pts': np.asarray(num_points)}),
                    'device': DataArray(np.asarray(device_times), dims=('pts',), coords={'pts': np.asarray(num_points)})})
