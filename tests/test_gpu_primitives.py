"""GPU parity tests for the reference's array primitives (csrc/xp_primitives.hpp through the C ABI): insert_level,
find_intersections, trapz, trap_around_zeros, bound_pressure, get_layer, shift_out_nans, from_most_unstable_parcel,
mix_layer, interp1d_numba, add_lcl_to_profile.

Oracle: oracle/parcel_oracle.py, the one-column NumPy restatement of the reference's where / shift / concat expressions
(pinned by the reference's KATs through the functions built on it; test_insert_level and test_parcel_profile_lcl call two
of the primitives directly and run through the C ABI in tests/test_gpu_parity.py::test_kat_through_c_abi).
Tolerances: the kernels evaluate the reference's expressions in its operation order without FMA contraction, so values
agree to the last bit of the library exp / log: 1e-12 relative here; NaN patterns, masks and integer results exactly;
sums 1e-12 relative (NumPy's pairwise summation order differs).
"""
import numpy as np
import pytest

from oracle import parcel_oracle as po
from xarray_parcel_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def xa():
    import torch
    assert torch.cuda.is_available(), 'these tests need the GPU'
    from xarray_parcel_amd import numpy_api
    return numpy_api


def _close(got, ref, rtol=1e-12, atol=0.0, what=''):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), (what, 'NaN pattern', np.argwhere(np.isnan(got) != np.isnan(ref))[:5])
    ok = ~np.isnan(ref)
    if ok.any():
        err = np.abs(got[ok] - ref[ok])
        lim = atol + rtol * np.abs(ref[ok])
        assert np.all(err <= lim), (what, float(err.max()), got[ok][np.argmax(err - lim)], ref[ok][np.argmax(err - lim)])


def _columns(nlev=24, ncol=300, seed=3, nan_fraction=0.08):
    return synth.columns(nlev=nlev, ncol=ncol, seed=seed, nan_fraction=nan_fraction, dtype=np.float64)


def _per_column(fn, ncol):
    """Run the one-column oracle over the columns and stack dict results along a new last axis."""
    rows = [fn(c) for c in range(ncol)]
    if isinstance(rows[0], dict):
        return {k: np.stack([np.asarray(r[k], dtype=np.float64) for r in rows], axis=-1) for k in rows[0]}
    return np.stack([np.asarray(r, dtype=np.float64) for r in rows], axis=-1)


def test_insert_level_vs_oracle(xa):
    p, t, td = _columns(seed=11)
    ncol = p.shape[1]
    rng = np.random.default_rng(1)
    lev_p = rng.uniform(150.0, 1050.0, ncol)
    lev_p[::7] = p[5, ::7]                                  # an existing coordinate: the old level stays below the new one
    lev_p[3::31] = np.nan                                   # NaN coordinate: the level lands in every row
    lev_p[5::29] = 1200.0                                   # below the surface
    lev_p[6::37] = 10.0                                     # above the top
    p[2, 4::13] = np.nan                                    # NaN coordinates inside the column (the fill-value trick)
    t[7, 5::11] = -999.0                                    # a value equal to the fill value comes out NaN
    lev_t, lev_td = rng.uniform(200, 300, ncol), rng.uniform(190, 290, ncol)
    got = xa.insert_level({'pressure': p, 'temperature': t, 'dewpoint': td},
                          {'pressure': lev_p, 'temperature': lev_t, 'dewpoint': lev_td}, coords='pressure')
    with np.errstate(invalid='ignore'):
        ref = _per_column(lambda c: po.insert_level({'pressure': p[:, c], 'temperature': t[:, c], 'dewpoint': td[:, c]},
                                                    {'pressure': lev_p[c], 'temperature': lev_t[c], 'dewpoint': lev_td[c]}), ncol)
    for k in ref:
        assert got[k].shape == (p.shape[0] + 1, ncol)
        _close(got[k], ref[k], rtol=0.0, what=k)           # pure data movement: exact
    # keys of `level` define the output (pf.py:983)
    assert set(xa.insert_level({'pressure': p, 'temperature': t}, {'pressure': lev_p}).keys()) == {'pressure'}


@pytest.mark.parametrize('log_x', [False, True])
def test_find_intersections_vs_oracle(xa, log_x):
    p, t, td = _columns(seed=12)
    ncol = p.shape[1]
    a = t + 3.0 * np.sin(np.arange(p.shape[0]))[:, None] + 0.05 * (t - td) - 0.4
    b = t.copy()
    a[6, ::9] = b[6, ::9]                                   # touching: sign 0 at a level
    got = xa.find_intersections(p, a, b, log_x=log_x)
    with np.errstate(all='ignore'):
        ref = _per_column(lambda c: po.find_intersections(p[:, c], a[:, c], b[:, c], log_x=log_x), ncol)
    assert set(got) == set(ref) and got['all_intersect_x'].shape == (p.shape[0] - 1, ncol)
    for k in ref:
        _close(got[k], ref[k], what=k)
    assert np.isfinite(got['all_intersect_x']).sum() > ncol  # the test data does cross
    # b = None stands for zero
    g0 = xa.find_intersections(p, a - b, None, log_x=log_x)
    for k in ('all_intersect_x', 'increasing_x', 'decreasing_x'):
        _close(g0[k], xa.find_intersections(p, a - b, np.zeros_like(p), log_x=log_x)[k], rtol=0.0, what=k)


def test_trapz_vs_oracle(xa):
    p, t, td = _columns(seed=13)
    nlev, ncol = p.shape
    y = t - td - 8.0                                        # both signs
    rng = np.random.default_rng(2)
    mask = rng.random((nlev - 1, ncol)) < 0.7
    for kw in (dict(), dict(only_positive=True), dict(only_negative=True)):
        for m in (None, mask):
            got = xa.trapz(y, np.log(p), mask=m, **kw)
            with np.errstate(invalid='ignore'):
                ref = _per_column(lambda c: po.trapz(y[:, c], np.log(p[:, c]), mask=None if m is None else m[:, c], **kw), ncol)
            _close(got, ref, rtol=1e-12, atol=1e-13, what=str(kw))
    d = xa.trapz({'a': y, 'b': t}, p)
    assert set(d) == {'a', 'b'} and d['a'].shape == (ncol,)
    with pytest.raises(AssertionError, match='Only negative OR positive'):
        xa.trapz(y, p, only_positive=True, only_negative=True)


@pytest.mark.parametrize('log_x', [True, False])
def test_trap_around_zeros_vs_oracle(xa, log_x):
    p, t, td = _columns(seed=14)
    nlev, ncol = p.shape
    y = t - td - 8.0 + 3.0 * np.cos(np.arange(nlev))[:, None]
    areas, mask = xa.trap_around_zeros(p, y, log_x=log_x)
    with np.errstate(all='ignore'):
        ref = [po.trap_around_zeros(p[:, c], y[:, c], log_x=log_x) for c in range(ncol)]
    for k in ('area', 'x', 'dx'):
        r = np.stack([a[k] for a, _ in ref], axis=-1)
        assert areas[k].shape == (2 * nlev - 1, ncol)
        _close(areas[k], r, what=k)
    _close(areas['x_from'], areas['x'] - areas['dx'] / 2, rtol=1e-15, what='x_from')       # pf.py:1276-1277
    _close(areas['x_to'], areas['x'] + areas['dx'] / 2, rtol=1e-15, what='x_to')
    rm = np.stack([m for _, m in ref], axis=-1)
    assert mask.shape == (nlev, ncol) and mask.dtype == bool
    assert np.array_equal(mask[:nlev - 1], rm) and mask[nlev - 1].all()
    assert (~mask).sum() > ncol // 2                        # zeros were found


def test_bound_pressure_and_get_layer_vs_oracle(xa):
    p, t, td = _columns(seed=15, nan_fraction=0.05)
    nlev, ncol = p.shape
    rng = np.random.default_rng(3)
    bound = rng.uniform(300.0, 1000.0, ncol)
    bound[::5] = 0.5 * (p[3, ::5] + p[4, ::5])              # exactly between two levels: the larger pressure wins
    _close(xa.bound_pressure(p, bound), _per_column(lambda c: po.bound_pressure(p[:, c], bound[c]), ncol), rtol=0.0)
    _close(xa.bound_pressure(p, 700.0), _per_column(lambda c: po.bound_pressure(p[:, c], 700.0), ncol), rtol=0.0)
    for interpolate in (True, False):
        for depth in (100.0, 300.0, 2000.0):
            got = xa.get_layer({'pressure': p, 'temperature': t, 'dewpoint': td}, depth=depth, interpolate=interpolate)
            with np.errstate(all='ignore'):
                ref = _per_column(lambda c: po.get_layer({'pressure': p[:, c], 'temperature': t[:, c], 'dewpoint': td[:, c]},
                                                         depth=depth, interpolate=interpolate), ncol)
            for k in ref:
                assert got[k].shape == (nlev + (1 if interpolate else 0), ncol)
                _close(got[k], ref[k], rtol=1e-13, what=(k, interpolate, depth))


def test_shift_out_nans_vs_reference_loop(xa):
    p, t, td = _columns(seed=16)
    nlev, ncol = p.shape
    lead = np.random.default_rng(4).integers(0, nlev + 1, ncol)
    for c in range(ncol):
        p[:lead[c], c] = np.nan
    got = xa.shift_out_nans({'pressure': p, 'temperature': t}, 'pressure')
    # the reference's loop (pf.py:1714-1718), restated on the whole grid
    x = {'pressure': p.copy(), 'temperature': t.copy()}
    for _ in range(nlev):
        first = np.isnan(x['pressure'][0])
        if not first.any():
            break
        x = {k: np.where(first[None, :], np.concatenate([v[1:], np.full((1, ncol), np.nan)]), v) for k, v in x.items()}
    for k in x:
        _close(got[k], x[k], rtol=0.0, what=k)


def _drop_all_nan_levels(cols):
    """dropna(dim, how='all') over the grid for a dict of (nlev, ncol) arrays."""
    alln = np.all(np.stack([np.isnan(v) for v in cols.values()]), axis=(0, 2))
    return {k: v[~alln] for k, v in cols.items()}, ~alln


@pytest.mark.parametrize('mode', ['most_unstable', 'mixed_layer'])
def test_rebased_profiles_vs_the_reference_expressions(xa, mode):
    """from_most_unstable_parcel (pf.py:1517) / mix_layer (pf.py:1604): where + dropna(how='all') + shift_out_nans
    (+ the parcel underneath), restated on the grid with the oracle's parcels."""
    p, t, td = _columns(nlev=30, ncol=200, seed=17, nan_fraction=0.0)
    nlev, ncol = p.shape
    if mode == 'most_unstable':
        # lift the lowest levels out of every column's search result: the grid-wide dropna then removes levels
        td[:3] -= 25.0
        rp, rt, rtd, parcel, kept = xa.from_most_unstable_parcel(p, t, td, depth=300)
        with np.errstate(all='ignore'):
            mus = [po.most_unstable_parcel(p[:, c], t[:, c], td[:, c], depth=300) for c in range(ncol)]
        thr = np.array([m['pressure'] for m in mus])
        keep = p <= thr[None, :]
        assert np.array_equal(np.asarray(parcel['index']), np.array([m['index'] for m in mus]))
    else:
        rp, rt, rtd, parcel, kept = xa.mix_layer(p, t, td, depth=100)
        with np.errstate(all='ignore'):
            mus = [po.mixed_parcel(p[:, c], t[:, c], td[:, c], depth=100) for c in range(ncol)]
        keep = p < (np.nanmax(p, axis=0) - 100.0)[None, :]
    for k in ('pressure', 'temperature', 'dewpoint'):
        _close(parcel[k], np.array([m[k] for m in mus]), rtol=1e-10, what='parcel ' + k)
    masked = {'pressure': np.where(keep, p, np.nan), 'temperature': np.where(keep, t, np.nan), 'dewpoint': np.where(keep, td, np.nan)}
    dropped, surv = _drop_all_nan_levels(masked)
    assert np.array_equal(kept, surv) and (~surv).any()      # some levels do get dropped
    n = dropped['pressure'].shape[0]
    for _ in range(n):                                       # shift_out_nans
        first = np.isnan(dropped['pressure'][0])
        if not first.any():
            break
        dropped = {k: np.where(first[None, :], np.concatenate([v[1:], np.full((1, ncol), np.nan)]), v) for k, v in dropped.items()}
    if mode == 'mixed_layer':
        dropped = {k: np.concatenate([np.asarray(parcel[k])[None, :], v]) for k, v in dropped.items()}
    for k, g in (('pressure', rp), ('temperature', rt), ('dewpoint', rtd)):
        _close(g, dropped[k], rtol=0.0, what=k)
    # the same through the column oracle's driver inputs (what cape_cin receives, padding aside)
    if mode == 'mixed_layer':
        for c in range(0, ncol, 23):
            op, ot, otd, _ = po.mix_layer(p[:, c], t[:, c], td[:, c], depth=100)
            m = len(op)
            _close(np.asarray(rp)[:m, c][~np.isnan(op)], op[~np.isnan(op)], rtol=1e-10, what='oracle mix_layer')


def test_interp1d_is_numpy_interp(xa):
    rng = np.random.default_rng(5)
    n, m, ncol = 40, 17, 250
    xp = np.sort(rng.uniform(0.0, 100.0, (n, ncol)), axis=0)
    fp = rng.normal(size=(n, ncol))
    at = rng.uniform(-10.0, 110.0, (m, ncol))
    at[3, ::7] = xp[10, ::7]                                 # on a knot
    at[4, ::9] = np.nan
    at[5, ::11] = xp[0, ::11]
    at[6, ::13] = xp[-1, ::13]
    got = xa.interp1d(at, xp, fp)
    ref = np.stack([np.interp(at[:, c], xp[:, c], fp[:, c]) for c in range(ncol)], axis=-1)
    _close(got, ref, rtol=1e-14, atol=1e-15)
    # one set of points for all columns (what the table lookup does with its pressure axis)
    got = xa.interp1d(at, xp[:, 0], fp)
    ref = np.stack([np.interp(at[:, c], xp[:, 0], fp[:, c]) for c in range(ncol)], axis=-1)
    _close(got, ref, rtol=1e-14, atol=1e-15)
    g32 = xa.interp1d(at.astype(np.float32), xp.astype(np.float32), fp.astype(np.float32))
    assert g32.dtype == np.float32


def test_add_lcl_to_profile_is_the_fused_profile(xa):
    """add_lcl_to_profile on parcel_profile's output = what parcel_profile_with_lcl's one pass returns (pf.py:806-856
    is exactly that composition)."""
    p, t, td = _columns(nlev=30, ncol=200, seed=18, nan_fraction=0.0)
    for interp in ('log', 'linear'):
        prof = xa.parcel_profile(p, p[0], t[0], td[0], moist='exact')
        env = {'pressure': p, 'temperature': t, 'dewpoint': td,
               'virtual_temperature': xa.virtual_temperature(t, xa.mixing_ratio(t, td, p))}
        got = xa.add_lcl_to_profile(prof, environment=env, interpolator=interp)
        ref = xa.parcel_profile_with_lcl(p, t, td, p[0], t[0], td[0], lcl_interp=interp, moist='exact')
        # columns whose LCL coincides with a level keep that level below the inserted one in both forms
        for k in ('pressure', 'temperature', 'virtual_temperature', 'environment_temperature', 'environment_dewpoint',
                  'environment_virtual_temperature'):
            _close(got[k], ref[k], rtol=1e-9, what=(interp, k))
    with pytest.raises(AssertionError, match='interpolator must be linear or log'):
        xa.add_lcl_to_profile(prof, environment=env, interpolator='cubic')


def test_device_tensors_and_fp32(xa):
    import torch
    p, t, td = _columns(seed=19, nan_fraction=0.0)
    dp, dt_, dtd = (torch.as_tensor(v.astype(np.float32)).cuda() for v in (p, t, td))
    r = xa.find_intersections(dp, dt_, dtd)
    assert r['all_intersect_x'].is_cuda and r['all_intersect_x'].dtype == torch.float32
    h = xa.find_intersections(p.astype(np.float32), t.astype(np.float32), td.astype(np.float32))
    assert np.array_equal(r['all_intersect_x'].cpu().numpy(), h['all_intersect_x'], equal_nan=True)
    lay = xa.get_layer({'pressure': dp, 'temperature': dt_}, depth=100)
    assert lay['temperature'].is_cuda and lay['temperature'].shape == (p.shape[0] + 1, p.shape[1])
    rp, rt, rtd, parcel, kept = xa.from_most_unstable_parcel(dp, dt_, dtd)
    assert rp.is_cuda and rp.shape[0] == int(kept.sum())
    a, m = xa.trap_around_zeros(dp, dt_ - dtd - 8.0)
    assert a['area'].is_cuda and m.dtype == torch.bool
    s = xa.trapz(dt_ - dtd - 8.0, torch.log(dp), mask=m[:-1])
    assert s.is_cuda and s.shape == (p.shape[1],)


# -- the xarray-facing mirror (reference signatures, Dataset / DataArray in and out) ---------------------------------------
def _da(values, dims, name=None, **coords):
    from xarray_parcel_amd._xr import DataArray
    return DataArray(np.asarray(values, dtype=np.float64), dims=dims, coords=coords, name=name)


def test_mirror_insert_level_is_the_references_test(xa):
    """unit_tests.py:1388-1411 verbatim through the mirror: insertion of a level containing an existing pressure."""
    from xarray_parcel_amd import parcel_functions as pf
    from xarray_parcel_amd._xr import Dataset
    lv = 'model_level_number'
    d = Dataset({'pressure': _da([[1000, 900, 800, 700], [1000, 900, 800, 700]], ('x', lv), x=[1, 2], **{lv: [1, 2, 3, 4]}),
                 'temperature': _da([[1, 1, 1, 1], [1, 1, 1, 1]], ('x', lv), x=[1, 2], **{lv: [1, 2, 3, 4]})})
    level = Dataset({'pressure': _da([1000, 600], ('x',), x=[1, 2]), 'temperature': _da([1.5, 2], ('x',), x=[1, 2])})
    res = pf.insert_level(d=d, level=level, coords='pressure')
    assert res['pressure'].dims == (lv, 'x')
    np.testing.assert_array_equal(res['pressure'].values.T, [[1000, 1000, 900, 800, 700], [1000, 900, 800, 700, 600]])
    np.testing.assert_array_equal(res['temperature'].values.T, [[1, 1.5, 1, 1, 1], [1, 1, 1, 1, 2]])
    np.testing.assert_array_equal(res['pressure'].coords[lv], [1, 2, 3, 4, 5])              # re-indexed (pf.py:971)
    bad = Dataset({'pressure': _da([[1000, -999, 800, 700]], ('x', lv), x=[1], **{lv: [1, 2, 3, 4]})})
    with pytest.raises(AssertionError, match='dataset d contains fill_value'):
        pf.insert_level(d=bad, level=Dataset({'pressure': _da([900], ('x',), x=[1])}), coords='pressure')


def test_mirror_primitives_shapes_labels_and_values(xa):
    from xarray_parcel_amd import parcel_functions as pf
    from xarray_parcel_amd._xr import Dataset
    lv = 'model_level_number'
    p, t, td = _columns(nlev=20, ncol=12, seed=21, nan_fraction=0.0)
    nlev = p.shape[0]
    g = lambda v, name: _da(v.reshape(nlev, 3, 4), (lv, 'y', 'x'), name=name, y=[0, 1, 2], x=[0, 1, 2, 3], **{lv: np.arange(nlev) + 5})
    P, T, TD = g(p, 'pressure'), g(t, 'temperature'), g(td, 'dewpoint')
    dat = Dataset({'pressure': P, 'temperature': T, 'dewpoint': TD})

    fi = pf.find_intersections(x=P, a=T, b=g(t - 0.5 * (t - td) + np.sin(np.arange(nlev))[:, None], 'b'), dim=lv, log_x=True)
    assert fi['all_intersect_x'].dims == ('offset_dim', 'y', 'x') and fi['all_intersect_x'].shape == (nlev - 1, 3, 4)
    np.testing.assert_array_equal(fi['all_intersect_x'].coords['offset_dim'], np.arange(nlev - 1) + 6)

    lay = pf.get_layer(dat, depth=100)
    assert lay['temperature'].shape == (nlev + 1, 3, 4)
    ref = xa.get_layer({'pressure': p, 'temperature': t, 'dewpoint': td}, depth=100)
    np.testing.assert_array_equal(lay['temperature'].values.reshape(nlev + 1, -1), ref['temperature'])
    lay0 = pf.get_layer(dat, depth=300, interpolate=False)
    assert lay0['pressure'].shape == (nlev, 3, 4)

    bp = pf.bound_pressure(P, bound=_da(np.full((3, 4), 700.0), ('y', 'x')), vert_dim=lv)
    assert bp.dims == ('y', 'x')
    np.testing.assert_array_equal(bp.values.reshape(-1), xa.bound_pressure(p, 700.0))

    tz = pf.trapz(dat, x='pressure', dim=lv)
    assert set(tz.keys()) == {'pressure', 'temperature', 'dewpoint'} and tz['temperature'].dims == ('y', 'x')
    np.testing.assert_allclose(tz['temperature'].values.reshape(-1), xa.trapz(t, p), rtol=0, atol=0)

    areas, mask = pf.trap_around_zeros(x=P, y=g(t - td - 8.0, 'y'), dim=lv)
    assert areas['area'].shape == (2 * nlev - 1, 3, 4) and mask.shape == (nlev, 3, 4)
    np.testing.assert_array_equal(areas['area'].coords[lv], np.concatenate([np.arange(nlev) + 5, np.arange(1, nlev) + 5]))
    tz2 = pf.trapz(Dataset({'d': g(t - td - 8.0, 'd'), 'lp': g(np.log(p), 'lp')}), x='lp', dim=lv, mask=mask)   # n-row mask (pf.py:1361)
    np.testing.assert_array_equal(tz2['d'].values.reshape(-1), xa.trapz(t - td - 8.0, np.log(p), mask=mask.values.reshape(nlev, -1)[:-1]))

    sh = pf.shift_out_nans(x=Dataset({'pressure': g(np.where(np.arange(nlev)[:, None] < 3, np.nan, p), 'pressure'), 'temperature': T}),
                           name='pressure', dim=lv)
    np.testing.assert_array_equal(sh['temperature'].values[:nlev - 3].reshape(nlev - 3, -1), t[3:])
    assert np.isnan(sh['pressure'].values[nlev - 3:]).all()

    rp, rt, rtd, layer = pf.from_most_unstable_parcel(P, T, TD, vert_dim=lv, depth=300)
    n = rp.shape[0]
    assert rp.dims == (lv, 'y', 'x') and set(layer.keys()) == {'pressure', 'temperature', 'dewpoint'}
    assert np.array_equal(rp.coords[lv], (np.arange(nlev) + 5)[nlev - n:])                  # dropna keeps the labels
    np.testing.assert_array_equal(rp.values[0].reshape(-1)[np.isfinite(layer['pressure'].values.reshape(-1))],
                                  layer['pressure'].values.reshape(-1)[np.isfinite(layer['pressure'].values.reshape(-1))])
    with pytest.raises(AssertionError, match='Pressure requires name pressure'):
        pf.from_most_unstable_parcel(g(p, 'p'), T, TD, vert_dim=lv)

    mp_, mt, mtd, mp = pf.mix_layer(P, T, TD, vert_dim=lv, depth=100)
    assert mp_.coords[lv][0] == mp_.coords[lv][1] - 1                                       # pf.py:1641
    np.testing.assert_array_equal(mp_.values[0], mp['pressure'].values)
    ref_mp = pf.mixed_parcel(P, T, TD, depth=100, vert_dim=lv)
    np.testing.assert_array_equal(mp['temperature'].values, ref_mp['temperature'].values)

    at = np.linspace(50.0, 1100.0, 7)
    out = pf.interp1d_numba(np.broadcast_to(at, (3, 4, 7)), p[::-1].T.reshape(3, 4, nlev), t[::-1].T.reshape(3, 4, nlev))
    assert out.shape == (3, 4, 7)
    np.testing.assert_allclose(out[1, 2], np.interp(at, p[::-1, 6], t[::-1, 6]), rtol=1e-14)
    assert pf.round_to(1.2345, 0.02) == 1.24 and pf.round_to(273.149, 0.5, dp=1) == 273.0   # pf.py:358


def test_mirror_add_lcl_to_profile_kat(xa):
    """unit_tests.py:205-230 (test_parcel_profile_lcl) with the reference's own call sequence: parcel_profile, then
    add_lcl_to_profile with a temperature-only environment, linear interpolation."""
    from tests import kat_recipes as kr
    from xarray_parcel_amd import parcel_functions as pf
    from xarray_parcel_amd._xr import Dataset
    lv = 'model_level_number'
    i = kr.inputs('test_parcel_profile_lcl')
    n = len(i['p'])
    p = _da(i['p'], (lv,), name='pressure', **{lv: np.arange(n)})
    t = _da(i['t'], (lv,), name='temperature', **{lv: np.arange(n)})
    prof = pf.parcel_profile(pressure=p, parcel_pressure=i['parcel_pressure'], parcel_temperature=i['parcel_temperature'],
                             parcel_dewpoint=i['parcel_dewpoint'], moist='exact')
    prof = pf.add_lcl_to_profile(profile=prof, environment=Dataset({'temperature': t, 'pressure': prof['pressure']}),
                                 interpolator='linear')
    kr.check('test_parcel_profile_lcl', {'prof.pressure': prof['pressure'].values, 'prof.environment_temperature':
                                         prof['environment_temperature'].values, 'prof.temperature': prof['temperature'].values})
    assert prof['pressure'].attrs['long_name'] == 'Pressure at LCL' and prof['pressure'].shape == (n + 1,)
