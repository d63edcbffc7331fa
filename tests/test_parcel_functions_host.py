"""Host logic of the xarray-facing mirror (xarray_parcel_amd/parcel_functions.py): DataArray plumbing,
attrs, prefixes, assert messages, MU/ML profile trimming.  CPU part: the kernel call is replaced by a
stand-in built on the oracle (there is no CPU product path); GPU part: the same calls for real."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests import kat_recipes as kr
from xarray_parcel_amd import parcel_functions as pf
from xarray_parcel_amd._xr import DataArray, Dataset

VD = 'model_level_number'


def _da(x, name=None):
    x = np.asarray(x, dtype=np.float64)
    return DataArray(x, dims=(VD,), coords={VD: np.arange(1, len(x) + 1)}, attrs={'units': 'K'}, name=name)


def _stand_in(p, t, td, parcel='surface', depth=None, parcel_values=None, want_profile=False, want=None, moist=None,
              **kw):
    sh = p.shape[1:]
    f = lambda a: np.asarray(a, dtype=np.float64).reshape(p.shape[0], -1)
    pv = None if parcel_values is None else np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64).reshape(-1),
                                                                      (f(p).shape[1],)) for v in parcel_values])
    r = co.cape_cin_grid(f(p), f(t), f(td), parcel=parcel, depth=depth, parcel_values=pv, want_profile=True,
                         moist='rk4', **kw)
    out = {k: (v.reshape(sh) if k != 'profile' else {kk: vv.reshape((vv.shape[0],) + sh) for kk, vv in v.items()})
           for k, v in r.items()}
    for k in ('parcel_pressure', 'parcel_temperature', 'parcel_dewpoint'):
        out[k] = np.full(sh, np.nan)
    return out


@pytest.fixture
def stub(monkeypatch):
    monkeypatch.setattr(pf._api, 'cape_cin_columns', _stand_in)


def test_surface_based_structure_attrs_prefix(stub):
    i = kr.inputs('test_surface_based_cape_cin')
    p, t, td = _da(i['levels']), _da(i['temperatures']), _da(i['dewpoints'])
    res, prof = pf.surface_based_cape_cin(p, t, td)
    assert abs(float(res.cape.values) - 230.1982) < 0.015 and abs(float(res.cin.values) + 58.0673) < 0.015
    assert res.cape.attrs['units'] == 'J kg$^{-1}$' and res.cape.attrs['description'] == 'CAPE for surface-based parcel.'
    assert res.attrs['correction'].startswith('Virtual temperature correction used')
    assert prof.pressure.dims == (VD,) and prof.pressure.shape == (len(i['levels']) + 1,)
    assert set(('lfc_pressure', 'el_pressure', 'lcl_pressure', 'environment_virtual_temperature')) <= set(prof.keys())
    res2, _ = pf.surface_based_cape_cin(p, t, td, prefix='sb', virtual_temperature_correction=False,
                                        lcl_interp='linear')
    assert 'sb_cape' in res2 and abs(float(res2.sb_cape.values) - 75.0535) < 0.015
    assert res2.attrs['correction'].startswith('Virtual temperature correction not used')


def test_grid_dims_are_preserved(stub):
    from xarray_parcel_amd import synth
    p, t, td = synth.columns(20, 12, seed=2)
    mk = lambda a: DataArray(a.reshape(20, 3, 4).transpose(1, 0, 2), dims=('lat', VD, 'lon'),
                             coords={'lat': [1., 2., 3.], VD: np.arange(20), 'lon': [5., 6., 7., 8.]})
    res, prof = pf.surface_based_cape_cin(mk(p), mk(t), mk(td))
    assert res.cape.dims == ('lat', 'lon') and res.cape.shape == (3, 4)
    assert prof.temperature.dims == (VD, 'lat', 'lon') and prof.temperature.shape == (21, 3, 4)
    ref = co.cape_cin_grid(p, t, td, moist='rk4')
    assert np.allclose(res.cape.values.reshape(-1), ref['cape'])


def test_reference_asserts(stub):
    i = kr.inputs('test_surface_based_cape_cin')
    p, t, td = _da(i['levels']), _da(i['temperatures']), _da(i['dewpoints'])
    with pytest.raises(AssertionError, match='Pressure requires name pressure.'):
        pf.most_unstable_cape_cin(p, t, td)
    bad = DataArray(i['levels'], dims=(VD,), coords={VD: np.arange(len(i['levels'])) * 2})
    with pytest.raises(AssertionError, match='increments must all be 1'):
        pf.surface_based_cape_cin(bad, t, td)


def test_mu_profile_is_trimmed(stub):
    i = kr.inputs('test_most_unstable_parcel')
    res, prof, layer = pf.most_unstable_cape_cin(_da(i['levels'], 'pressure'), _da(i['temperatures'], 'temperature'),
                                                 _da(i['dewpoints'], 'dewpoint'), depth=100)
    assert prof.pressure.shape[0] == 3            # MU level 1 of 3 -> 2 levels + LCL
    assert res.cape.attrs['description'] == 'CAPE for most-unstable parcel in lowest 100 hPa.'


def test_default_moist_mode_is_the_references(stub, monkeypatch):
    """pf.py:525-607 + 56-61: an unchanged reference call sequence gets the lookup-table moist lapse, and a moist call
    before load_moist_adiabat_lookups() raises the reference's assert."""
    i = kr.inputs('test_surface_based_cape_cin')
    p, t, td = _da(i['levels']), _da(i['temperatures']), _da(i['dewpoints'])
    pf.set_moist_lapse(None)                                   # the module as a caller of the reference finds it
    loaded = {'v': 0}

    class _Lib:
        def xp_tables_loaded(self):
            return loaded['v']
    from xarray_parcel_amd import _lib
    monkeypatch.setattr(_lib, 'load', lambda: _Lib())
    for call in (lambda: pf.surface_based_cape_cin(p, t, td),
                 lambda: pf.moist_lapse(p, np.array([300.0])),
                 lambda: pf.parcel_profile(p, p[0:1], t[0:1], td[0:1]),
                 lambda: pf.wet_bulb_temperature(p, t, td),
                 lambda: pf.parcel_profile_with_lcl(p, t, td, p[0:1], t[0:1], td[0:1])):
        with pytest.raises(AssertionError, match='Call load_moist_adiabat_lookups first.'):
            call()
    seen = {}

    def spy(p_, t_, td_, **kw):
        seen.update(kw)
        kw.pop('moist')
        return _stand_in(p_, t_, td_, **kw)
    monkeypatch.setattr(pf._api, 'cape_cin_columns', spy)
    loaded['v'] = 1                                              # load_moist_adiabat_lookups() has run
    pf.surface_based_cape_cin(p, t, td)
    assert seen['moist'] == 'table'
    pf.surface_based_cape_cin(p, t, td, moist='family')          # opt-in per call
    assert seen['moist'] == 'family'
    pf.set_moist_lapse('exact')                                  # opt-in per module (what the KAT runs do)
    pf.surface_based_cape_cin(p, t, td)
    assert seen['moist'] == 'exact'


@pytest.mark.gpu
def test_default_moist_mode_on_gpu():
    """The same on the device: before the tables exist the reference's assert, after load_moist_adiabat_lookups() the
    lookup-table answer (= moist='table', and different from the ODE's by the table's error)."""
    i = kr.inputs('test_surface_based_cape_cin')
    p, t, td = _da(i['levels']), _da(i['temperatures']), _da(i['dewpoints'])
    from xarray_parcel_amd import _lib
    pf.set_moist_lapse(None)
    if not _lib.load().xp_tables_loaded():
        with pytest.raises(AssertionError, match='Call load_moist_adiabat_lookups first.'):
            pf.surface_based_cape_cin(p, t, td)
    pf.load_moist_adiabat_lookups(cache=False)
    res, _ = pf.surface_based_cape_cin(p, t, td)
    tab, _ = pf.surface_based_cape_cin(p, t, td, moist='table')
    ode, _ = pf.surface_based_cape_cin(p, t, td, moist='exact')
    assert float(res.cape.values) == float(tab.cape.values)
    assert 1e-3 < abs(float(res.cape.values) - float(ode.cape.values)) < 40.0     # demo.ipynb:320-329: up to 33 J/kg
    lev = _da([1000., 900., 800.])
    ml = np.asarray(pf.moist_lapse(lev, np.array([293.0])).values).ravel()
    assert np.array_equal(ml, np.asarray(pf.moist_lapse(lev, np.array([293.0]), moist='table').values).ravel())
    ode = np.asarray(pf.moist_lapse(lev, np.array([293.0]), moist='exact').values).ravel()
    assert 1e-4 < np.max(np.abs(ml - ode)) < 0.08                                 # the table's error (demo.ipynb:252: 0.037 K; :322: 0.077 K)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['test_surface_based_cape_cin', 'test_surface_based_cape_cin_mp',
                                  'test_sensitive_sounding', 'test_cape_cin_value_error'])
def test_mirror_on_gpu(name):
    i = kr.inputs(name)
    kw = kr.MP if name.endswith('_mp') or name == 'test_cape_cin_value_error' else {}
    res, prof = pf.surface_based_cape_cin(_da(i['levels']), _da(i['temperatures']), _da(i['dewpoints']), **kw)
    exp = {w: v for w, v, _ in kr.expected(name)}
    assert abs(float(res.cape.values) - exp['cape_cin.cape']) < 0.015
    assert abs(float(res.cin.values) - exp['cape_cin.cin']) < 0.015


@pytest.mark.gpu
def test_mirror_mu_ml_on_gpu():
    i = kr.inputs('test_mixed_layer_cape_cin')
    args = (_da(i['levels'], 'pressure'), _da(i['temperatures'], 'temperature'), _da(i['dewpoints'], 'dewpoint'))
    res, prof, mp = pf.mixed_layer_cape_cin(*args, **kr.MP)
    assert abs(float(res.cape.values) - 1096.7461) < 0.015 and abs(float(res.cin.values) + 20.6727) < 0.015
    assert abs(float(mp.pressure.values) - i['levels'][0]) < 1e-9
    res, prof, layer = pf.most_unstable_cape_cin(*args, depth=300)
    assert np.isfinite(float(res.cape.values))
    lay = pf.most_unstable_parcel(Dataset(dict(pressure=args[0], temperature=args[1], dewpoint=args[2])))
    assert float(lay.pressure.values) == float(layer.pressure.values)
