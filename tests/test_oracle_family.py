"""The "adiabat family" exact mode (oracle/family.py): accuracy against the ODE, C oracle vs NumPy spelling, KATs."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import family as fam
from oracle import parcel_oracle as po
from oracle import thermo as th
from tests import kat_recipes as kr
from tests.test_oracle_kat import RK4_LOOSEN


def test_family_is_within_5e6_K_of_the_ode():
    rng = np.random.default_rng(0)
    worst, handled = 0.0, 0
    for _ in range(120):
        p_l, t_l = rng.uniform(500, 1040), rng.uniform(235, 304)
        if th.saturation_vapor_pressure(t_l) > 0.15 * p_l:
            continue
        ps = np.sort(rng.uniform(40, p_l, 12))[::-1]
        a = fam.moist_lapse_family(ps, t_l, p_l)
        if np.isnan(fam.label(fam.table(), np.log(p_l), t_l)):
            assert np.array_equal(a, th.moist_lapse_rk4(ps, t_l, p_l))      # label outside the table: RK4 mode
            continue
        handled += 1
        b = th.moist_lapse_ode(ps, t_l, p_l, method='DOP853', atol=1e-13, rtol=1e-13)
        worst = max(worst, float(np.abs(a - b).max()))
    assert handled > 80 and worst < 5e-6, (handled, worst)


def test_c_and_numpy_tables_and_lookups_agree():
    tab_c = co.family_table()
    tab_py = fam.table()
    assert tab_c.shape == tab_py.shape and float(np.max(np.abs(tab_c - tab_py))) < 1e-10
    rng = np.random.default_rng(1)
    co.set_moist_lapse('family')
    try:
        for _ in range(60):
            p_l, t_l = rng.uniform(400, 1100), rng.uniform(200, 310)
            ps = np.sort(rng.uniform(20, 1150, 10))[::-1]           # also outside the table -> RK4 fall-back on both sides
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
