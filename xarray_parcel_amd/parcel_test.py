"""
Counterpart of the timing harness of traupach/xarray_parcel's modules/parcel_test.py ("pt.py") for the MI355X path:
time_function (pt.py:18-35), surface_cape_vector (pt.py:250-274) and benchmark_cape (pt.py:586-619).

The reference's `surface_cape_serial` leg (MetPy, one column at a time) has no counterpart here -- MetPy is not part of
this build -- so benchmark_cape reports the two legs that exist: 'xr_load' (host arrays in, host arrays out: includes
the PCIe copies, like the reference's timed region includes materialisation) and 'device' (inputs already resident
in HBM as torch tensors).
"""
import time

import numpy as np

from . import numpy_api as _api
from . import parcel_functions as parcel
from ._xr import DataArray, Dataset


def time_function(func, dat, **kwargs):
    """pt.py:18-35: run func(dat) and return (result, seconds); results are host arrays already (no .load())."""
    start = time.perf_counter()
    ret = func(dat, **kwargs)
    if hasattr(ret, 'load'):
        ret = ret.load()
    end = time.perf_counter()
    return ret, end - start


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
