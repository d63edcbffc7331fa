"""BASELINE.json configs c1, c3, c4, c5 as GPU parity tests: the true level count / dtype / outputs of each config at a
reduced column count, every column compared with the C oracle (tests/test_gpu_parity.py::_compare rules: CAPE / CIN
within fp32 output rounding of the oracle's fp64 result, LFC / EL / parcel indices and status words bit-exact).
c2 at full size is tests/test_gpu_parity.py::test_full_size_properties_config2.

Reference analogue: the differential test of modules/parcel_test.py:549-575 (vector path vs per-column path on 225
columns of test_data.nc) and benchmark_cape (:586-619).  test_data.nc itself is absent from the reference mount
(.MISSING_LARGE_BLOBS), so c1 runs on the synthetic stand-in of the same shape.
"""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import thermo as th
from tests.test_gpu_parity import _compare
from xarray_parcel_amd import synth

pytestmark = pytest.mark.gpu

xa = None


@pytest.fixture(scope='module', autouse=True)
def _api():
    global xa
    import torch
    assert torch.cuda.is_available(), 'these tests need the GPU'
    from xarray_parcel_amd import numpy_api
    xa = numpy_api
    yield


def _compare_profile(got, ref, dtype):
    """The (nlev + 1)-row profile: identical NaN pattern, values within 1e-8 (fp64) or fp32 rounding of the oracle's."""
    for k in ref:
        a, b = np.asarray(got[k], dtype=np.float64), ref[k]
        if dtype == np.float32:
            b = b.astype(np.float32).astype(np.float64)
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        ok = ~np.isnan(b)
        tol = 1e-8 if dtype == np.float64 else 2e-7 * np.maximum(np.abs(b[ok]), 1.0) + 1e-6
        err = np.abs(a[ok] - b[ok])
        assert np.all(err <= tol), (k, float(err.max()))


@pytest.mark.parametrize('moist', ['exact', 'family'])
@pytest.mark.parametrize('parcel', ['surface', 'most_unstable', 'mixed_layer'])
def test_config3_fp32_full_profile(parcel, moist):
    """c3: 128 levels, fp32, parcel profile + LCL / LFC / EL + CAPE / CIN (k_cape_cin<float, *, PROFILE=true>)."""
    nlev, ncol = 128, 6000
    p, t, td = synth.columns(nlev=nlev, ncol=ncol, seed=20250720, nan_fraction=0.05, dtype=np.float32)
    got = xa.cape_cin_columns(p, t, td, parcel=parcel, want_profile=True, moist=moist)
    ref = co.cape_cin_grid(p, t, td, parcel=parcel, moist='rk4' if moist == 'exact' else 'family', want_profile=True)
    _compare(got, ref, np.float32, 1e-6)
    _compare_profile(got['profile'], ref['profile'], np.float32)


@pytest.mark.parametrize('moist', ['exact', 'family'])
def test_config4_fp32_128_levels_surface(moist):
    """c4 (one rank's kernel): 128 levels, fp32, surface-based CAPE / CIN only."""
    nlev, ncol = 128, 20000
    p, t, td = synth.columns(nlev=nlev, ncol=ncol, seed=20250721, nan_fraction=0.02, dtype=np.float32)
    got = xa.cape_cin_columns(p, t, td, moist=moist)
    ref = co.cape_cin_grid(p, t, td, moist='rk4' if moist == 'exact' else 'family')
    _compare(got, ref, np.float32, 1e-6)


@pytest.mark.parametrize('moist', ['exact', 'family'])
@pytest.mark.parametrize('parcel', ['most_unstable', 'mixed_layer'])
def test_config5_time_flattened_mu_ml(parcel, moist):
    """c5: (time, lev, y, x) = (24, 100, ny, nx) fp32, most-unstable and mixed-layer parcels; the time axis is
    flattened into the column axis ((lev, time * y * x), what a rank of the 8-GPU run receives)."""
    import torch
    nt, nlev, ny, nx = 24, 100, 8, 40
    p, t, td = synth.columns(nlev=nlev, ncol=nt * ny * nx, seed=20250722, nan_fraction=0.03, dtype=np.float32)
    grid = [torch.from_numpy(a.reshape(nlev, nt, ny, nx)).cuda() for a in (p, t, td)]       # (lev, time, y, x)
    got = xa.cape_cin_columns(*grid, parcel=parcel, moist=moist)
    assert got['cape'].shape == (nt, ny, nx)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy().reshape(-1) for k, v in got.items()}
    ref = co.cape_cin_grid(p, t, td, parcel=parcel, moist='rk4' if moist == 'exact' else 'family')
    _compare(got, ref, np.float32, 1e-6)


def test_config1_stand_in_through_the_harness():
    """c1: the reference's benchmark body surface_cape_vector (parcel_test.py:250-274: q -> Td, then surface-based
    CAPE / CIN) on the 90 x 101 x 101 fp32 stand-in for test_data.nc, every column against the oracle fed the MetPy-1.4.1
    q -> Td chain (parity unpinned for that conversion); fused and two-step routes agree."""
    from xarray_parcel_amd import parcel_test as pt
    from xarray_parcel_amd._xr import DataArray, Dataset
    nlev, ny, nx = 90, 101, 101
    p, t, td = synth.columns(nlev, ny * nx, seed=20250718, dtype=np.float64)
    e = th.saturation_vapor_pressure(td)
    w = th.EPSILON * e / (p - e)
    q = w / (1.0 + w)
    p, t, q = (a.astype(np.float32) for a in (p, t, q))
    dims = ('model_level_number', 'latitude', 'longitude')

    def mk(a, n):
        return DataArray(a.reshape(nlev, ny, nx), dims=dims, name=n,
                         coords={'model_level_number': np.arange(nlev), 'latitude': np.arange(ny), 'longitude': np.arange(nx)})
    dat = Dataset({'pressure': mk(p, 'pressure'), 'temperature': mk(t, 'temperature'),
                   'specific_humidity': mk(q, 'specific_humidity')})
    fused = pt.surface_cape_vector(dat)
    two_step = pt.surface_cape_vector(dat, fused=False)
    with np.errstate(all='ignore'):
        tdr = th.dewpoint_from_specific_humidity(p.astype(np.float64), t.astype(np.float64), q.astype(np.float64))
    ref = co.cape_cin_grid(p.astype(np.float64), t.astype(np.float64), tdr, moist='rk4')
    for name, out in (('fused', fused), ('two_step', two_step)):
        for k in ('cape', 'cin'):
            a = np.asarray(out[k].values, dtype=np.float64).ravel()
            b = ref[k].astype(np.float32).astype(np.float64)
            # the two-step route rounds the dewpoint to fp32 between the kernels: CAPE moves by up to ~1e-2 J/kg
            tol = (2e-7 * np.maximum(np.abs(b), 1.0) + 1e-6) if name == 'fused' else 0.05
            assert np.all(np.abs(a - b) <= tol), (name, k, float(np.max(np.abs(a - b))))
    r = xa.cape_cin_columns(p, t, q, humidity='specific')
    _compare(r, ref, np.float32, 1e-6)


def test_self_test_leg_of_the_harness_vs_oracle(capsys):
    """conv_properties_xarray (parcel_test.py:416-547), the vectorised leg of the reference's self-test: every variable
    it returns against the one-column oracle on a small fp64 grid (exact moist mode -- what the reference's
    comparison with MetPy is run with); compare / compare_results (pt.py:37-66, 577-584) report nothing for equal
    sets and a line for a perturbed one."""
    from oracle import parcel_oracle as po
    from xarray_parcel_amd import parcel_functions as pf
    from xarray_parcel_amd import parcel_test as pt
    from xarray_parcel_amd._xr import DataArray, Dataset
    nlev, ny, nx = 40, 3, 4
    lv = 'model_level_number'
    p, t, td = synth.columns(nlev, ny * nx, seed=77, dtype=np.float64)
    e = th.saturation_vapor_pressure(td)
    w = th.EPSILON * e / (p - e)
    q = w / (1.0 + w)

    def mk(a, n):
        return DataArray(a.reshape(nlev, ny, nx), dims=(lv, 'latitude', 'longitude'), name=n,
                         coords={lv: np.arange(nlev), 'latitude': np.arange(ny), 'longitude': np.arange(nx)})
    dat = Dataset({'pressure': mk(p, 'pressure'), 'temperature': mk(t, 'temperature'), 'specific_humidity': mk(q, 'specific_humidity')})
    pf.set_moist_lapse('exact')
    out = pt.conv_properties_xarray(dat, virt_temp=True, lcl_interp='log', pos_cape_neg_cin=True)
    want = {'dewpoint', 'mp_pressure', 'mp_temperature', 'mp_dewpoint', 'dry_lapse_temp', 'moist_lapse_temp', 'mixed_cape', 'mixed_cin',
            'max_cape', 'max_cin', 'surf_pres', 'surface_profile', 'surf_temp', 'surface_lcl_pressure', 'surface_lcl_temp',
            'surface_lfc_pressure', 'surface_lfc_temp', 'surface_el_pressure', 'surface_el_temp', 'surface_cape', 'surface_cin',
            'lifted_index', 'dci', 'wet_bulb_temperature', 'wet_bulb_temperature_fast'}
    assert want <= set(out.keys()), want - set(out.keys())
    assert out['surface_profile'].dims[0] == lv + '_lcl' and out['surface_profile'].shape == (nlev + 1, ny, nx)
    with np.errstate(all='ignore'):
        tdr = th.dewpoint_from_specific_humidity(p, t, q)
    flat = lambda k: np.asarray(out[k].values, dtype=np.float64).reshape(out[k].shape[0], -1) if out[k].ndim == 3 else \
        np.asarray(out[k].values, dtype=np.float64).reshape(-1)
    assert np.max(np.abs(flat('dewpoint') - tdr)) <= 1e-10
    po.set_moist_lapse('rk4')
    try:
        per_col = []
        for c in range(ny * nx):
            with np.errstate(all='ignore'):
                sb, sprof = po.surface_based_cape_cin(p[:, c], t[:, c], tdr[:, c])
                ml, mprof, mp = po.mixed_layer_cape_cin(p[:, c], t[:, c], tdr[:, c], depth=100)
                mu, _, _ = po.most_unstable_cape_cin(p[:, c], t[:, c], tdr[:, c], depth=300)
                le = po.lfc_el(sprof['pressure'], sprof['temperature'], sprof['environment_temperature'], sprof['lcl_pressure'],
                               sprof['lcl_temperature'])
                li = po.lifted_index(mprof)
            per_col.append((sb, sprof, ml, mp, mu, le, li))
    finally:
        po.set_moist_lapse('ode')
    for c in range(ny * nx):
        sb, sprof, ml, mp, mu, le, li = per_col[c]
        for k, v in (('surface_cape', sb['cape']), ('surface_cin', sb['cin']), ('mixed_cape', ml['cape']), ('mixed_cin', ml['cin']),
                     ('max_cape', mu['cape']), ('max_cin', mu['cin']), ('mp_temperature', mp['temperature']), ('mp_dewpoint', mp['dewpoint']),
                     ('surface_lcl_pressure', sprof['lcl_pressure']), ('surface_lfc_pressure', le['lfc_pressure']),
                     ('surface_el_pressure', le['el_pressure']), ('lifted_index', li)):
            a = flat(k)[c]
            assert (np.isnan(a) and np.isnan(v)) or abs(a - v) <= 1e-6 * max(1.0, abs(v)), (k, c, a, v)
        a, b = flat('surface_profile')[:, c], np.asarray(sprof['temperature'])
        assert np.allclose(a[:len(b)], b, rtol=0, atol=1e-6, equal_nan=True)
    capsys.readouterr()
    pt.compare_results(out, out)
    assert 'K' not in capsys.readouterr().out.split('\n', 1)[1]               # header only
    worse = Dataset({'surface_cape': DataArray(out['surface_cape'].values + 1.0, dims=out['surface_cape'].dims, attrs=out['surface_cape'].attrs)})
    assert pt.compare(worse['surface_cape'], out['surface_cape'], name='surface_cape') is False
    assert 'surface_cape' in capsys.readouterr().out
