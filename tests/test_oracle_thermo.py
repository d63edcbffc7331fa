"""Properties of the oracle's thermodynamics (oracle/thermo.py) that the parity claims lean on."""
import numpy as np
import pytest

from oracle import thermo as th


@pytest.mark.parametrize('p0,t0', [(950., 296.), (900., 285.), (700., 270.), (990., 303.), (600., 250.)])
@pytest.mark.parametrize('nlev', [32, 64, 128])
def test_rk4_spec_is_within_2e5_K_of_the_ode(p0, t0, nlev):
    """The build's exact-mode specification (RK4 in ln p, steps <= 0.1) against a tight DOP853 solve of MetPy's
    pseudo-adiabat: <= 2e-5 K on sigma-like level sets from the LCL to 60 hPa (MetPy's own LSODA tolerance gives
    4e-5 ... 4e-4 K)."""
    sig = 1 - (np.arange(nlev) / (nlev - 1)) ** 1.3
    p = 60 + (p0 - 60) * sig
    spec = th.moist_lapse_rk4(p, t0, p0)
    ref = th.moist_lapse_ode(p, t0, p0, method='DOP853', atol=1e-12, rtol=1e-12)
    assert np.max(np.abs(spec - ref)) < 2e-5


def test_rk4_spec_handles_both_directions_and_nans():
    p = np.array([1050., np.nan, 800., 1000., 600.])
    out = th.moist_lapse_rk4(p, 293., 1000.)
    ref = th.moist_lapse_ode(np.array([1050., 800., 1000., 600.]), 293., 1000., method='DOP853', atol=1e-12, rtol=1e-12)
    assert np.isnan(out[1]) and out[3] == 293.
    assert np.max(np.abs(out[[0, 2, 3, 4]] - ref)) < 2e-5


def test_per_column_lcl_agrees_with_metpy_fixed_point():
    """Steffensen per column (the build's LCL) vs SciPy's fixed_point on the scalar (what a one-column KAT sees):
    identical iteration, so identical to rounding; LCL snaps onto the parcel level when saturated."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        p, t = rng.uniform(700, 1040), rng.uniform(250, 310)
        td = t - rng.uniform(0, 25)
        a = th.lcl_steffensen(p, t, td)
        b = th.lcl_metpy(p, t, td)
        assert abs(a[0] - float(b[0])) <= 1e-9 * p and abs(a[1] - float(b[1])) <= 1e-9
    p_l, t_l, _ = th.lcl_steffensen(1000., 290., 290.)
    assert p_l == 1000.


def test_lcl_16_digit_kat():
    """unit_tests.py:258-270 (test_lcl_nans, disabled upstream because of MetPy's block-wide stop rule)."""
    p_l, t_l, _ = th.lcl_steffensen(900., 25. + 273.15, 20. + 273.15)
    assert abs(p_l - 836.4098648012595) < 1e-9 and abs(t_l - (18.82281982535794 + 273.15)) < 1e-9
