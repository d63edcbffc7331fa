"""GPU parity tests for the SURVEY 8(f) rows built on the hot path: the q -> dewpoint front step (stand-alone and fused
into xp_cape_cin), the single-level indices (pf.py:1830-1870, 2102-2214) and the harness counterpart (parcel_test.py).

Oracle: oracle/thermo.py + oracle/parcel_oracle.py (NumPy restatement, one column at a time).  None of these functions
has a KAT in the reference's unit tests except lifted_index, so for them parity is against the restatement only
("parity unpinned" for dewpoint_from_specific_humidity).
Tolerances: element-wise fp64 formulas 1e-10 K; interpolated / intersected values 1e-9 relative; CAPE / CIN through the
fused path 1e-6 J/kg with bit-exact level indices.
"""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import parcel_oracle as po
from oracle import thermo as th
from tests.test_gpu_parity import _compare
from xarray_parcel_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def xa():
    import torch
    assert torch.cuda.is_available(), 'these tests need the GPU'
    from xarray_parcel_amd import numpy_api
    return numpy_api


def _specific_humidity(p, td):
    """Specific humidity of air with dewpoint td (exact thermodynamics; MetPy 1.4.1's chain maps it back to within
    ~0.1 K of td, see tests/test_oracle_indices.py)."""
    e = th.saturation_vapor_pressure(td)
    w = th.EPSILON * e / (p - e)
    return w / (1.0 + w)


def _heights(p):
    return 44330.8 * (1.0 - (p / 1013.25) ** 0.190263)       # standard-atmosphere altitude [m], monotone in p


def test_dewpoint_from_specific_humidity(xa):
    p, t, td = synth.columns(nlev=30, ncol=500, seed=5, nan_fraction=0.05, dtype=np.float64)
    q = _specific_humidity(p, td)
    q[3, ::17] = 0.0                                                # log(0): NaN in MetPy's formula chain
    q[4, ::19] = -1e-4
    got = xa.dewpoint_from_specific_humidity(p, t, q)
    with np.errstate(all='ignore'):
        ref = th.dewpoint_from_specific_humidity(p, t, q)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.max(np.abs(got[ok] - ref[ok])) <= 1e-10
    g32 = xa.dewpoint_from_specific_humidity(p.astype(np.float32), t.astype(np.float32), q.astype(np.float32))
    assert g32.dtype == np.float32
    with np.errstate(all='ignore'):
        r32 = th.dewpoint_from_specific_humidity(p.astype(np.float32).astype(np.float64), t.astype(np.float32).astype(np.float64),
                                                 q.astype(np.float32).astype(np.float64))
    ok = ~np.isnan(r32)
    assert np.array_equal(np.isnan(g32), ~ok)
    assert np.max(np.abs(g32[ok] - r32[ok])) <= 3e-5


def test_mixing_ratio_and_virtual_temperature(xa):
    """pf.py:684 / pf.py:782 element-wise, against the oracle's MetPy 1.4.1 forms (bit-for-bit up to libm: 1e-15 relative)."""
    p, t, td = synth.columns(nlev=20, ncol=300, seed=61, nan_fraction=0.05, dtype=np.float64)
    w = xa.mixing_ratio(t, td, p)
    with np.errstate(all='ignore'):
        ref = po.mixing_ratio(t, td, p)
    assert w.shape == t.shape and np.array_equal(np.isnan(w), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.max(np.abs(w[ok] / ref[ok] - 1.0)) <= 1e-14
    tv = xa.virtual_temperature(t, w)
    assert np.nanmax(np.abs(tv - po.virtual_temperature(t, ref))) <= 1e-10
    from xarray_parcel_amd import parcel_functions as pf
    from xarray_parcel_amd._xr import DataArray
    W = pf.mixing_ratio(DataArray(t, dims=('z', 'c')), DataArray(td, dims=('z', 'c')), DataArray(p, dims=('z', 'c')))
    assert W.attrs['units'] == 'kg kg$^{-1}$' and np.allclose(W.values, w, equal_nan=True)
    assert pf.virtual_temperature(DataArray(t, dims=('z', 'c')), W).attrs['long_name'] == 'Virtual temperature'


@pytest.mark.parametrize('parcel', ['surface', 'most_unstable', 'mixed_layer'])
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_fused_specific_humidity_input(xa, parcel, dtype):
    """XP_HUM_SPECIFIC: the moisture view holds q and is converted on load; oracle = convert, then the same driver."""
    p, t, td = synth.columns(nlev=48, ncol=8000, seed=23, nan_fraction=0.08, dtype=np.float64)
    q = _specific_humidity(p, td).astype(dtype)
    p, t = p.astype(dtype), t.astype(dtype)
    with np.errstate(all='ignore'):
        td_ref = th.dewpoint_from_specific_humidity(p.astype(np.float64), t.astype(np.float64), q.astype(np.float64))
    got = xa.cape_cin_columns(p, t, q, parcel=parcel, humidity='specific')
    ref = co.cape_cin_grid(p.astype(np.float64), t.astype(np.float64), td_ref, parcel=parcel, moist='rk4')
    _compare(got, ref, dtype, 1e-6)
    # and the two-step route of the reference (parcel_test.py:262-271) gives the same answer as the fused one
    two = xa.cape_cin_columns(p, t, xa.dewpoint_from_specific_humidity(p, t, q), parcel=parcel)
    if dtype == np.float64:
        same = np.asarray(two['lfc_index']) == np.asarray(got['lfc_index'])
        assert same.mean() > 0.999
        a, b = np.asarray(two['cape'])[same], np.asarray(got['cape'])[same]
        assert np.nanmax(np.abs(a - b)) <= 1e-6


def test_single_level_indices_vs_oracle(xa):
    p, t, td = synth.columns(nlev=40, ncol=300, seed=31, nan_fraction=0.06, dtype=np.float64)
    z = _heights(p)
    # make some columns cross 273.15 K several times / never / exactly on a level
    t2 = t.copy()
    t2[5:9, ::7] += 12.0 * np.sin(np.arange(4))[:, None]
    t2[:, 1::11] = np.minimum(t2[:, 1::11], 270.0)                   # never reaches freezing: NaN
    t2[6, 2::13] = 273.15                                           # a level exactly on the isotherm
    flh = xa.freezing_level_height(t2, z)
    mlh, wb = xa.melting_level_height(p, t2, td, z)
    lr = xa.lapse_rate(p, t, z)
    t500 = xa.isobar_temperature(p, t, 500.0)
    res = xa.cape_cin_columns(p, t, td, parcel='mixed_layer', want_profile=True)
    li = xa.lifted_index(res['profile'])
    dci = xa.deep_convective_index(p, t, td, li)

    def close(a, b, what, c, tol=1e-9):
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= tol * max(1.0, abs(b)), (what, c, a, b)

    with np.errstate(all='ignore'):
        for c in range(p.shape[1]):
            close(flh[c], po.freezing_level_height(t2[:, c], z[:, c]), 'flh', c)
            r_mlh, r_wb = po.melting_level_height(p[:, c], t2[:, c], td[:, c], z[:, c])
            close(mlh[c], r_mlh, 'mlh', c)
            close(lr[c], po.lapse_rate(p[:, c], t[:, c], z[:, c]), 'lapse', c)
            close(t500[c], po.isobar_temperature(p[:, c], t[:, c], 500.0), 't500', c)
            r_li = po.lifted_index({k: res['profile'][k][:, c] for k in ('pressure', 'temperature', 'environment_temperature')})
            close(li[c], r_li, 'li', c)
            close(dci[c], po.deep_convective_index(p[:, c], t[:, c], td[:, c], r_li), 'dci', c, tol=1e-8)
    cold = np.setdiff1d(np.arange(1, p.shape[1], 11), np.arange(2, p.shape[1], 13))
    assert np.isnan(flh[cold]).all() and np.isfinite(flh[::7]).any()
    # exact wet bulb leg of melting_level_height (fast=False) runs the Normand kernel
    mlh2, wb2 = xa.melting_level_height(p[:, :16], t2[:, :16], td[:, :16], z[:, :16], fast=False)
    with np.errstate(all='ignore'):
        for c in range(0, 16, 5):
            po.set_moist_lapse('rk4')
            try:
                r, _ = po.melting_level_height(p[:, c], t2[:, c], td[:, c], z[:, c], fast=False)
            finally:
                po.set_moist_lapse('ode')
            close(mlh2[c], r, 'mlh exact', c, tol=1e-7)


def test_xarray_mirrors_and_harness(xa):
    """parcel_functions / parcel_test mirrors: names, attrs and values (against the array API they wrap)."""
    from xarray_parcel_amd import parcel_functions as pf
    from xarray_parcel_amd import parcel_test as pt
    from xarray_parcel_amd._xr import DataArray, Dataset
    nlev, ny, nx = 30, 9, 11
    p, t, td = synth.columns(nlev=nlev, ncol=ny * nx, seed=41, dtype=np.float64)
    q = _specific_humidity(p, td)
    dims = ('model_level_number', 'latitude', 'longitude')
    coords = {'model_level_number': np.arange(nlev), 'latitude': np.linspace(-30, -20, ny), 'longitude': np.linspace(140, 150, nx)}
    mk = lambda a, n: DataArray(a.reshape(nlev, ny, nx), dims=dims, coords=coords, name=n)
    P, T, TD, Q, Z = mk(p, 'pressure'), mk(t, 'temperature'), mk(td, 'dewpoint'), mk(q, 'specific_humidity'), mk(_heights(p), 'height')
    flh = pf.freezing_level_height(temperature=T, height=Z)
    assert flh.name == 'freezing_level' and flh.attrs['units'] == 'm' and flh.dims == ('latitude', 'longitude')
    assert np.allclose(flh.values.ravel(), xa.freezing_level_height(t, _heights(p)), equal_nan=True)
    mlh, wb = pf.melting_level_height(pressure=P, temperature=T, dewpoint=TD, height=Z)
    assert mlh.name == 'melting_level' and wb.attrs['description'] == 'Estimated using 1/3 method.'
    lr = pf.lapse_rate(pressure=P, temperature=T, height=Z)
    assert lr.attrs['description'] == '700-500 hPa lapse rate' and lr.attrs['units'] == 'K km$^{-1}$'
    t500 = pf.isobar_temperature(pressure=P, temperature=T, isobar=500)
    assert t500.attrs == {'description': 'Temperature at 500 hPa.', 'long_name': 'Isobar temperature', 'units': 'K'}
    cc, prof, _ = pf.mixed_layer_cape_cin(pressure=P, temperature=T, dewpoint=TD, prefix='mixed_100')
    li = pf.lifted_index(profile=prof, prefix='mixed_100', description='x')
    dci = pf.deep_convective_index(pressure=P, temperature=T, dewpoint=TD, lifted_index=li['mixed_100_lifted_index'], prefix='mixed_100')
    assert 'mixed_100_dci' in dci and dci['mixed_100_dci'].attrs['units'] == 'C'
    tdq = pf.dewpoint_from_specific_humidity(pressure=P, temperature=T, specific_humidity=Q)
    assert tdq.dims == dims and np.nanmax(np.abs(tdq.values.reshape(nlev, -1) - th.dewpoint_from_specific_humidity(p, t, q))) < 1e-9
    # harness: fused and two-step routes agree; benchmark_cape returns one time per sub-grid
    dat = Dataset({'pressure': P, 'temperature': T, 'specific_humidity': Q})
    a, b = pt.surface_cape_vector(dat), pt.surface_cape_vector(dat, fused=False)
    assert np.nanmax(np.abs(a['cape'].values - b['cape'].values)) <= 1e-6
    ref = co.cape_cin_grid(p, t, th.dewpoint_from_specific_humidity(p, t, q), moist='rk4')
    assert np.nanmax(np.abs(a['cape'].values.ravel() - ref['cape'])) <= 1e-6
    bench = pt.benchmark_cape(dat, points=[2, 4, 9])
    assert list(bench['xr_load'].coords['pts']) == [4, 16, 81] and np.all(bench['device'].values > 0)


def test_interp_levels_equals_interp_level(xa):
    """xp_interp_levels: several variables at several coordinates in one pass = xp_interp_level one at a time, bit for bit
    (NaN levels, duplicate coordinates, coordinates outside the column included)."""
    p, t, td = synth.columns(nlev=30, ncol=777, seed=61, nan_fraction=0.1, dtype=np.float64)
    z = _heights(p)
    p2 = p.copy(); p2[7] = p2[6]                                           # a duplicated coordinate
    for dtype in (np.float64, np.float32):
        for log in (True, False):
            cds = p2.astype(dtype)
            xs = [t.astype(dtype), td.astype(dtype), z.astype(dtype)]
            ats = [850.0, 700.0, 500.0, 30.0]
            got = xa.interp_levels(cds, xs, ats, log=log)
            for v, x in enumerate(xs):
                for j, at in enumerate(ats):
                    assert np.array_equal(got[v][j], xa.interp_level(cds, x, at, log=log), equal_nan=True), (dtype, log, v, at)
    one = xa.interp_levels(p, [t], [500.0], log=True)
    assert np.array_equal(one[0][0], xa.isobar_temperature(p, t, 500.0), equal_nan=True)


def _bundle_inputs(nlev=40, ncol=48, seed=51, nan_fraction=0.06):
    p, t, td = synth.columns(nlev=nlev, ncol=ncol, seed=seed, nan_fraction=nan_fraction, dtype=np.float64)
    q = _specific_humidity(p, td)
    z = _heights(p)
    rng = np.random.default_rng(seed)
    nw = 12
    wh = np.linspace(50.0, 9000.0, nw)[:, None] + rng.uniform(0, 40, (1, ncol))        # wind heights above the surface [m]
    wu = 5.0 + wh * 2.5e-3 + rng.normal(0, 3, (nw, ncol))
    wv = -2.0 + wh * 1.0e-3 + rng.normal(0, 3, (nw, ncol))
    return {'pressure': p, 'temperature': t, 'specific_humidity': q, 'height_asl': z, 'wind_u': wu, 'wind_v': wv,
            'wind_height_above_surface': wh, 'surface_wind_u': rng.normal(2, 2, ncol), 'surface_wind_v': rng.normal(0, 2, ncol)}


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_conv_properties_one_call_equals_the_composition(xa, dtype):
    """xp_conv_properties (one pass over the four grids + three parcel passes + one per-point kernel) against the
    composition of stand-alone calls and array arithmetic it replaces: identical NaN pattern, values to rounding of the
    per-point arithmetic (the parcel results and the interpolated values are bit-identical); host arrays and device
    tensors; ignore_nans."""
    import torch
    d = {k: np.asarray(v, dtype=dtype) for k, v in _bundle_inputs(nlev=33, ncol=3000, seed=7).items()}
    dd = {k: torch.as_tensor(v).cuda() for k, v in d.items()}          # (the composition keeps the data's type only for device tensors)
    rtol = 1e-12 if dtype == np.float64 else 2e-6
    for moist in ('exact', 'family'):
        for ignore in (False, True):
            ref = xa.conv_properties_composed(dd, ignore_nans=ignore, moist=moist)
            got = xa.conv_properties(d, ignore_nans=ignore, moist=moist)
            assert set(got) == set(ref)
            for k in ref:
                a, b = np.asarray(got[k]), ref[k].cpu().numpy()
                if k == 'positive_shear':
                    assert a.dtype == bool and np.array_equal(a, b.astype(bool)), (moist, ignore, k)
                    continue
                assert a.dtype == dtype and np.array_equal(np.isnan(a), np.isnan(b)), (moist, ignore, k)
                ok = ~np.isnan(b)
                err = np.abs(a[ok].astype(np.float64) - b[ok].astype(np.float64))
                # (float32: the composition forms differences of ~300 K numbers in float32 -- an ulp there is 3e-5 K --, the
                # per-point kernel in float64)
                atol = 0.0 if dtype == np.float64 else 3e-4
                assert np.all(err <= atol + rtol * np.maximum(1.0, np.abs(b[ok]))), (moist, ignore, k, float(err.max()))
                if k.endswith('_cape') or k.endswith('_cin') or k in ('temp_500', 'freezing_level', 'melting_level'):
                    assert np.array_equal(a[ok], b[ok]), (moist, ignore, k)
    dev = xa.conv_properties(dd)
    host = xa.conv_properties(d)
    for k in host:
        assert np.array_equal(dev[k].cpu().numpy(), host[k], equal_nan=True), k


def test_conv_properties_and_storm_proxies_vs_oracle(xa):
    """The reference's product bundle (pf.py:1951 conv_properties, pf.py:2323 storm_proxies) as compositions of the
    device calls, against the same compositions of the oracle, column by column."""
    d = _bundle_inputs()
    got = xa.conv_properties(d)
    ncol = d['pressure'].shape[1]
    po.set_moist_lapse('rk4')
    try:
        with np.errstate(all='ignore'):
            ref = [po.conv_properties(*(d[k][:, c] for k in ('pressure', 'temperature', 'specific_humidity', 'height_asl')),
                                      d['surface_wind_u'][c], d['surface_wind_v'][c], d['wind_u'][:, c], d['wind_v'][:, c],
                                      d['wind_height_above_surface'][:, c]) for c in range(ncol)]
    finally:
        po.set_moist_lapse('ode')
    tol = {'cape': 1e-6, 'cin': 1e-6}
    for k in got:
        g = np.asarray(got[k])
        r = np.array([x[k] for x in ref])
        if k == 'positive_shear':
            assert np.array_equal(g.astype(bool), r.astype(bool)), k
            continue
        assert np.array_equal(np.isnan(g), np.isnan(r)), (k, np.nonzero(np.isnan(g) != np.isnan(r))[0][:5])
        ok = ~np.isnan(r)
        err = np.abs(g[ok] - r[ok])
        lim = tol.get(k.split('_')[-1], 1e-8) * np.maximum(1.0, np.abs(r[ok]))
        assert np.all(err <= lim), (k, float(err.max()))
    assert np.isnan(np.asarray(got['mu_cape'])).sum() >= 1            # NaN columns are blanked (pf.py:2097-2098)
    gp, rp = xa.storm_proxies(got), po.storm_proxies({k: np.array([x[k] for x in ref]) for k in ref[0]})
    for k in rp:
        if k == 'ship':
            a, b = np.asarray(gp[k]), rp[k]
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.nanmax(np.abs(a - b), initial=0.0) <= 1e-9
        else:
            assert np.array_equal(np.asarray(gp[k]).astype(bool), rp[k].astype(bool)), k
    # the minimal bundle (pf.py:1873) on the NaN-free columns
    clean = np.nonzero(~np.isnan(np.asarray(got['mu_cape'])))[0][:12]
    dm = {k: (v[..., clean] if np.ndim(v) else v) for k, v in d.items()}
    gm = xa.min_conv_properties(dm)
    po.set_moist_lapse('rk4')
    try:
        with np.errstate(all='ignore'):
            for i, c in enumerate(clean):
                rm = po.min_conv_properties(*(d[k][:, c] for k in ('pressure', 'temperature', 'specific_humidity', 'height_asl')),
                                            d['surface_wind_u'][c], d['surface_wind_v'][c], d['wind_u'][:, c], d['wind_v'][:, c],
                                            d['wind_height_above_surface'][:, c])
                for k in rm:
                    a_, b_ = float(np.asarray(gm[k])[i]), float(rm[k])
                    assert (np.isnan(a_) and np.isnan(b_)) or abs(a_ - b_) <= 1e-6 * max(1.0, abs(b_)), (k, c, a_, b_)
    finally:
        po.set_moist_lapse('ode')
    # the xarray-facing mirror: same numbers, reference names / attrs
    from xarray_parcel_amd import parcel_functions as pf
    from xarray_parcel_amd._xr import DataArray, Dataset
    nlev, nw = d['pressure'].shape[0], d['wind_u'].shape[0]
    ds = Dataset({k: DataArray(d[k], dims=('model_level_number', 'point')) for k in ('pressure', 'temperature', 'specific_humidity', 'height_asl')})
    for k in ('wind_u', 'wind_v', 'wind_height_above_surface'):
        ds[k] = DataArray(d[k], dims=('wind_level', 'point'))
    for k in ('surface_wind_u', 'surface_wind_v'):
        ds[k] = DataArray(d[k], dims=('point',))
    props = pf.conv_properties(ds)
    assert props['mixed_100_dci'].attrs['units'] == 'C' and props['mu_cape'].dims == ('point',)
    assert np.allclose(props['mu_cape'].values, np.asarray(got['mu_cape']), equal_nan=True)
    assert set(pf.min_conv_properties(ds).keys()) >= {'mixed_100_cape', 'temp_500', 'shear_magnitude'}
    prox = pf.storm_proxies(props)
    assert prox['proxy_Kunz2007'].attrs['long_name'] == 'Proxy Kunz 2007'
    assert np.array_equal(prox['proxy_SHIP_0.1'].values.astype(bool), np.asarray(gp['proxy_SHIP_0.1']).astype(bool))


def test_per_point_product_kernels_vs_oracle(xa):
    """xp_storm_proxies / xp_significant_hail_parameter / xp_wind_shear (pf.py:2323, 2261, 2216) on inputs that straddle
    every threshold and validity window (the bundle's own outputs on the synthetic columns rarely do), NaNs included:
    flags identical, SHIP and shear to the last bits (same operation order, no FMA contraction)."""
    import torch
    rng = np.random.default_rng(9)
    n = 4000
    d = {'mu_cape': rng.uniform(-200, 4000, n), 'mu_mixing_ratio': rng.uniform(0.008, 0.016, n),
         'mixed_100_cape': rng.uniform(-100, 3000, n), 'mixed_100_cin': rng.uniform(-120, 5, n),
         'mixed_100_lifted_index': rng.uniform(-8, 6, n), 'mixed_100_dci': rng.uniform(5, 40, n),
         'mixed_50_cape': rng.uniform(-100, 3000, n), 'mixed_50_cin': rng.uniform(-60, 5, n),
         'lapse_rate_700_500': rng.uniform(-9, -4, n), 'temp_500': rng.uniform(245, 275, n),
         'freezing_level': rng.uniform(500, 5000, n), 'shear_magnitude': rng.uniform(0, 40, n),
         'positive_shear': rng.random(n) < 0.6}
    for i, k in enumerate(k for k in d if k != 'positive_shear'):
        d[k][i::53] = np.nan
    got = xa.storm_proxies(d)
    ref = po.storm_proxies(d)
    assert list(got) == list(ref)                                      # the reference's order of variables
    for k in ref:
        if k == 'ship':
            assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k])) and np.isfinite(ref[k]).sum() > 100
            assert np.nanmax(np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e-300)) <= 4e-16
        else:
            assert got[k].dtype == bool and np.array_equal(got[k], ref[k]), k
            assert 0 < ref[k].sum() < n, k                             # both outcomes occur
    ship = xa.significant_hail_parameter(d['mu_cape'], d['mu_mixing_ratio'], d['lapse_rate_700_500'], d['temp_500'],
                                         d['shear_magnitude'], d['freezing_level'])
    with np.errstate(invalid='ignore'):
        r = po.significant_hail_parameter(d['mu_cape'], d['mu_mixing_ratio'], d['lapse_rate_700_500'], d['temp_500'],
                                          d['shear_magnitude'], d['freezing_level'])
    assert np.array_equal(np.isnan(ship), np.isnan(r)) and np.nanmax(np.abs(ship - r) / np.maximum(np.abs(r), 1e-300)) <= 4e-16
    # device tensors, fp32
    dt_ = {k: torch.as_tensor(v.astype(np.float32) if v.dtype != bool else v).cuda() for k, v in d.items()}
    g32 = xa.storm_proxies(dt_)
    assert g32['ship'].is_cuda and g32['ship'].dtype == torch.float32 and g32['proxy_Craven2004'].dtype == torch.bool
    r32 = po.storm_proxies({k: (v.astype(np.float32).astype(np.float64) if v.dtype != bool else v) for k, v in d.items()})
    assert (g32['proxy_Marsh2009'].cpu().numpy() != r32['proxy_Marsh2009']).sum() == 0
    # wind shear: (nwind, ncol) winds on their own heights
    nw, ncol = 9, 500
    h = np.sort(rng.uniform(10, 9000, (nw, ncol)), axis=0)
    h[3, ::7] = 6000.0                                                  # a level exactly at the shear height
    u, v = rng.normal(0, 12, (nw, ncol)), rng.normal(0, 12, (nw, ncol))
    u[2, ::11] = np.nan
    su, sv = rng.normal(0, 5, ncol), rng.normal(0, 5, ncol)
    gs = xa.wind_shear(su, sv, u, v, h)
    with np.errstate(invalid='ignore'):
        rs = [po.wind_shear(su[c], sv[c], u[:, c], v[:, c], h[:, c]) for c in range(ncol)]
    for k in ('shear_u', 'shear_v', 'shear_magnitude'):
        r = np.array([x[k] for x in rs])
        assert np.array_equal(np.isnan(gs[k]), np.isnan(r)) and np.nanmax(np.abs(gs[k] - r)) <= 1e-12, k
    assert np.array_equal(gs['positive_shear'], np.array([x['positive_shear'] for x in rs]))
    assert gs['positive_shear'].dtype == bool
