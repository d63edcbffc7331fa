"""The C oracle (oracle/c/xp_oracle.c) against (a) the reference's KATs and (b) the Python oracle
on seeded synthetic columns incl. NaNs.  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import parcel_oracle as po
from tests import kat_recipes as kr
from tests.test_oracle_kat import RK4_LOOSEN
from xarray_parcel_amd import synth


@pytest.mark.parametrize('name', kr.applicable(co))
def test_kat_c_oracle(name):
    co.set_moist_lapse('rk4')
    kr.run(name, co, loosen=RK4_LOOSEN.get(name))


MODES = [dict(), dict(virtual_temperature_correction=False, lcl_interp='linear'),
         dict(pos_cape_neg_cin=False), dict(post_zero_cin=True, lcl_interp='linear')]


@pytest.mark.parametrize('parcel', ['surface', 'most_unstable', 'mixed_layer'])
@pytest.mark.parametrize('mode', range(len(MODES)))
def test_c_vs_python_oracle(parcel, mode):
    kw = MODES[mode]
    p, t, td = synth.columns(nlev=40, ncol=60, seed=100 + mode, nan_fraction=0.1, dtype=np.float64)
    got = co.cape_cin_grid(p, t, td, parcel=parcel, moist='rk4', **kw)
    fn = {'surface': po.surface_based_cape_cin, 'most_unstable': po.most_unstable_cape_cin,
          'mixed_layer': po.mixed_layer_cape_cin}[parcel]
    po.set_moist_lapse('rk4')
    try:
        for c in range(p.shape[1]):
            res = fn(p[:, c], t[:, c], td[:, c], per_column_lcl=True, **kw)
            cc, prof = res[0], res[1]
            for k in ('cape', 'cin'):
                assert np.isclose(got[k][c], cc[k], rtol=1e-9, atol=1e-9), (c, k, got[k][c], cc[k])
            for k in ('lcl_pressure', 'lfc_pressure', 'lfc_temperature', 'el_pressure', 'el_temperature'):
                a, b = got[k][c], prof[k]
                assert (np.isnan(a) and np.isnan(b)) or np.isclose(a, b, rtol=1e-10, atol=1e-9), (c, k, a, b)
            assert got['lfc_index'][c] == prof['lfc_index'], (c, got['lfc_index'][c], prof['lfc_index'])
            assert got['el_index'][c] == prof['el_index'], (c, got['el_index'][c], prof['el_index'])
    finally:
        po.set_moist_lapse('ode')


# ---- reference lookup-table mode (pf.py:525-607) ------------------------------------------------------------
@pytest.fixture(scope='module')
def tables():
    from oracle import tables as tb
    tab = tb.get_tables()
    co.set_tables(tab)
    return tab


@pytest.mark.parametrize('name', kr.MOIST_LAPSE_KATS)
def test_table_mode_moist_lapse_kats(name, tables):
    """run_moist_lapse_tests_looser (unit_tests.py:106-112): the four moist-lapse KATs at 2 decimals, through
    the Python emulation and the C oracle."""
    po.set_moist_lapse('table', tables)
    try:
        kr.run(name, po, loosen=2)
    finally:
        po.set_moist_lapse('ode')
    co.set_moist_lapse('table')
    try:
        kr.run(name, co, loosen=2)
    finally:
        co.set_moist_lapse('rk4')


def test_table_statistics_match_the_published_figures(tables):
    """20 % of index cells are empty (plot at parcel_functions_demo.ipynb:221) and the table is within 0.037 K of
    the ODE over 1000 -> 100 hPa, T0 = 250 ... 313 K (parcel_functions_demo.ipynb:252)."""
    from oracle import tables as tb, thermo as th
    assert abs((tables.index == 0).mean() - 0.202) < 0.005
    worst = 0.0
    p = np.arange(1000., 100., -4.)
    for t0 in np.linspace(250, 313, 22):
        a = tb.moist_lapse_table(tables, p, t0, 1000.)
        b = th.moist_lapse_rk4(p, t0, 1000.)
        worst = max(worst, float(np.nanmax(np.abs(a - b))))
    assert 0.02 < worst < 0.0375, worst


def test_table_mode_c_vs_python(tables):
    p, t, td = synth.columns(nlev=40, ncol=40, seed=21, nan_fraction=0.1, dtype=np.float64)
    got = co.cape_cin_grid(p, t, td, moist='table')
    po.set_moist_lapse('table', tables)
    try:
        for c in range(p.shape[1]):
            cc, prof = po.surface_based_cape_cin(p[:, c], t[:, c], td[:, c], per_column_lcl=True)
            assert np.isclose(got['cape'][c], cc['cape'], rtol=1e-9, atol=1e-9), (c, got['cape'][c], cc['cape'])
            assert np.isclose(got['cin'][c], cc['cin'], rtol=1e-9, atol=1e-9)
            assert got['lfc_index'][c] == prof['lfc_index'] and got['el_index'][c] == prof['el_index']
    finally:
        po.set_moist_lapse('ode')


def test_float32_storage_of_the_adiabats_moves_cape_by_less_than_a_hundredth(tables):
    """The reference keeps its 14 300 adiabats in float64 (pf.py:507-511); this build stores them in float32 (126 MB
    instead of 251 MB; rounding <= 1.5e-5 K at 250 K).  Bound of the effect on the product: surface-based CAPE / CIN in
    table mode with the float32-stored rows against the same rows kept in float64, same index table."""
    from oracle import tables as tb
    from xarray_parcel_amd import synth
    ncol = 200
    p, t, td = synth.columns(nlev=48, ncol=ncol, seed=23, dtype=np.float64)
    # only the rows these columns select need float64 copies: solve them again (same DOP853 call as build_tables)
    levels, temps = tb.grids()
    starts = np.empty(2 * len(temps)); starts[0::2] = temps; starts[1::2] = temps + tb.T_STEP / 2
    out32, out64, rows = [], [], {}
    for c in range(ncol):
        po.set_moist_lapse('table', tables)
        a = po.surface_based_cape_cin(p[:, c], t[:, c], td[:, c])
        out32.append((a[0]['cape'], a[0]['cin']))
        lcl_p, lcl_t = a[1]['lcl_pressure'], a[1]['lcl_temperature']
        ip = tb.nearest_index_descending(lcl_p, tables.p_max, tables.p_step, tables.n_p)
        jt = tb.nearest_index_ascending(lcl_t, tables.t_min, tables.t_step, tables.n_t)
        rows[c] = int(tables.index[ip, jt])
    need = sorted({r for r in rows.values() if r > 0})
    prof = tb.solve_adiabats(starts[np.array(need) - 1], levels)[:, ::-1]          # float64, pressure ascending
    assert np.max(np.abs(prof.astype(np.float32).astype(np.float64) - tables.adiabats[np.array(need) - 1])) < 1e-4
    ad64 = tables.adiabats.astype(np.float64)                                      # float32 values everywhere ...
    ad64[np.array(need) - 1] = prof                                                # ... float64 where it matters here
    tab64 = tb.Tables(index=tables.index, adiabats=ad64)
    try:
        po.set_moist_lapse('table', tab64)
        for c in range(ncol):
            a = po.surface_based_cape_cin(p[:, c], t[:, c], td[:, c])
            out64.append((a[0]['cape'], a[0]['cin']))
    finally:
        po.set_moist_lapse('ode')
        po._MOIST.pop('tables', None)
    d = np.abs(np.array(out32) - np.array(out64))
    assert len(need) > 50 and d[:, 0].max() < 1e-2 and d[:, 1].max() < 1e-2, (len(need), d.max(axis=0))
