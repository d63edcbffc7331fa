"""The "adiabat family" exact mode (oracle/family.py): accuracy against the ODE, C oracle vs NumPy spelling, KATs."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import family as fam
from oracle import parcel_oracle as po
from oracle import thermo as th
from tests import kat_recipes as kr
from tests.test_oracle_kat import RK4_LOOSEN


def test_family_is_within_1e6_K_of_the_ode():
    """< 1e-6 K inside the table (1100 ... 20 hPa), < 3e-6 K with the dry continuation above it (down to 5 hPa)."""
    rng = np.random.default_rng(0)
    worst, worst_top, handled = 0.0, 0.0, 0
    for _ in range(160):
        p_l, t_l = rng.uniform(300, 1090), rng.uniform(225, 309)
        if th.saturation_vapor_pressure(t_l) > 0.15 * p_l:
            continue
        ps = np.sort(rng.uniform(5, p_l, 14))[::-1]
        a = fam.moist_lapse_family(ps, t_l, p_l)
        psi, q = fam.label(fam.table(), np.log(p_l), t_l, p_l)
        if np.isnan(psi):
            assert np.array_equal(a, th.moist_lapse_rk4(ps, t_l, p_l))      # label outside the table: RK4 mode
            continue
        assert abs(fam.evaluate(fam.table(), np.log(p_l), psi, q) - t_l) < 1e-10   # the curve passes through the LCL
        assert abs(fam.evaluate_tv(fam.table(), np.log(p_l), psi, q) - fam.virtual_temperature(p_l, t_l)) < 1e-10
        handled += 1
        b = th.moist_lapse_ode(ps, t_l, p_l, method='DOP853', atol=1e-13, rtol=1e-13)
        inside = np.log(ps) >= fam.XLO
        if inside.any():
            worst = max(worst, float(np.abs(a - b)[inside].max()))
        worst_top = max(worst_top, float(np.abs(a - b).max()))
    assert handled > 100 and worst < 1e-6 and worst_top < 3e-6, (handled, worst, worst_top)


def test_temperature_of_inverts_the_virtual_temperature():
    """The parcel temperature is DEFINED through the tabulated virtual temperature: five Newton steps recover T to
    1e-12 K wherever e_s <= 0.1 p (everything a label of the table can reach)."""
    rng = np.random.default_rng(4)
    worst = 0.0
    for _ in range(4000):
        p, t = rng.uniform(20, 1100), rng.uniform(150, 317)
        if th.saturation_vapor_pressure(t) > 0.10 * p:
            continue
        worst = max(worst, abs(float(fam.temperature_of(p, fam.virtual_temperature(p, t))) - t))
    assert worst < 1e-12, worst


def test_c_and_numpy_tables_and_lookups_agree():
    tab_c = co.family_table()
    tab_py = fam.table()
    # the high-order monomial coefficients are conditioned to ~1e-9 (long double vs double solve); the polynomials
    # they define agree to ~1e-13 K
    assert tab_c.shape == tab_py.shape and float(np.max(np.abs(tab_c - tab_py))) < 1e-8
    rng = np.random.default_rng(1)
    for _ in range(500):
        j, q, z, s = rng.integers(fam.NPX), rng.integers(fam.NPS), rng.uniform(-1, 1), rng.uniform(-1, 1)
        va, vb = (fam._horner(fam.column_poly(tab, j, q, s), z) for tab in (tab_c, tab_py))
        assert abs(va - vb) < 1e-11
    co.set_moist_lapse('family')
    try:
        for _ in range(60):
            p_l, t_l = rng.uniform(400, 1100), rng.uniform(200, 310)
            ps = np.sort(rng.uniform(5, 1150, 10))[::-1]            # also outside the table -> RK4 fall-back on both sides
            a = co.moist_lapse(ps, t_l, p_l)
            b = fam.moist_lapse_family(ps, t_l, p_l)
            assert np.allclose(a, b, rtol=0, atol=1e-9, equal_nan=True), (p_l, t_l, a, b)
    finally:
        co.set_moist_lapse('rk4')


@pytest.mark.parametrize('name', kr.applicable(co))
def test_kats_in_family_mode(name):
    """All KATs with the family mode in the C oracle (loosened exactly where the RK4 mode is)."""
    co.set_moist_lapse('family')
    try:
        kr.run(name, co, loosen=RK4_LOOSEN.get(name))
    finally:
        co.set_moist_lapse('rk4')
