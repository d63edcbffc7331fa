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


def test_fp32_ln_theta_e_ranks_like_fp64_outside_the_window():
    """The device's most-unstable search ranks the layer's levels by an fp32 ln(theta_e) (xp_kernels.hpp ln_theta_e_f32) and
    only repeats the search in fp64 when the two best levels are within MU_F32_WINDOW = 2e-5 of each other.  That is
    safe as long as the fp32 value is within half the window of the fp64 one: here the same formula in NumPy float32
    stays within 2e-6 on the synthetic soundings (the hardware's log2 / exp2 / rcp are 1-ulp instructions like NumPy's)."""
    from xarray_parcel_amd import synth
    f = np.float32
    p, t, td = synth.columns(100, 4000, seed=3, dtype=np.float64)
    pf, tf, tdf = f(p), f(t), f(td)
    e = f(6.112) * np.exp2((f(17.67) - f(4302.645) / (tdf - f(29.65))) * f(1.4426950408889634))
    r = f(0.6219569100577033) * e / (pf - e)
    l2t, l2td = np.log2(tf), np.log2(tdf)
    tl = f(56) + f(1) / (f(1) / (tdf - f(56)) + (l2t - l2td) * f(0.6931471805599453 / 800))
    got = (f(0.6931471805599453) * (l2t + f(2 / 7) * (f(np.log2(1000.0)) - np.log2(pf - e)) + f(0.28) * r * (l2t - np.log2(tl))) +
           r * (f(1) + f(0.448) * r) * (f(3036) / tl - f(1.78))).astype(np.float64)
    ref = np.log(th.equivalent_potential_temperature(p, t, td))
    layer = p > p[0] - 320.0
    assert np.abs(got - ref)[layer].max() < 2e-6
