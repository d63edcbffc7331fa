"""CPU tests of the oracle's restatement of the single-level indices (pf.py:1830-1870, 2102-2214) and of the harness
front step (parcel_test.py:262-266) on hand-checkable inputs.  The reference holds no KAT for these functions."""
import numpy as np

from oracle import parcel_oracle as po
from oracle import thermo as th


def test_freezing_level_is_the_lowest_crossing():
    z = np.array([0.0, 1000.0, 2000.0, 3000.0, 4000.0])
    t = np.array([283.15, 278.15, 268.15, 275.15, 263.15])       # crosses at 1500 m (down), 2714 m (up), 3583 m (down)
    assert abs(po.freezing_level_height(t, z) - 1500.0) < 1e-9
    assert np.isnan(po.freezing_level_height(t - 30.0, z))        # never reaches 0 C
    t2 = np.array([283.15, 273.15, 263.15, 253.15, 243.15])       # a level exactly on the isotherm
    assert abs(po.freezing_level_height(t2, z) - 1000.0) < 1e-9
    t3 = np.array([283.15, np.nan, 263.15, 253.15, 243.15])       # NaN gap hides the crossing (pf.py:1019-1046)
    assert np.isnan(po.freezing_level_height(t3, z))


def test_melting_level_uses_the_one_third_rule():
    z = np.array([0.0, 1000.0, 2000.0])
    t = np.array([285.15, 276.15, 267.15])
    td = np.array([279.15, 270.15, 261.15])
    wb = po.wet_bulb_temperature_fast(t, td)
    assert np.allclose(wb, t - 2.0)
    mlh, _ = po.melting_level_height(None, t, td, z)
    assert abs(mlh - (1000.0 + 1000.0 * (274.15 - 273.15) / 9.0)) < 1e-9


def test_lapse_rate_isobar_temperature_and_dci():
    p = np.array([1000.0, 850.0, 700.0, 500.0, 300.0])
    t = np.array([300.0, 290.0, 280.0, 260.0, 230.0])
    td = t - 5.0
    z = np.array([100.0, 1500.0, 3000.0, 5500.0, 9000.0])
    assert abs(po.lapse_rate(p, t, z) - (260.0 - 280.0) / (5.5 - 3.0)) < 1e-12
    assert po.isobar_temperature(p, t, 700.0) == 280.0
    mid = po.isobar_temperature(p, t, 600.0)                       # log-p interpolation between 700 and 500 hPa
    assert abs(mid - (280.0 + (260.0 - 280.0) * (np.log(600 / 700) / np.log(500 / 700)))) < 1e-12
    assert abs(po.deep_convective_index(p, t, td, -3.0) - ((290.0 - 273.15) + (285.0 - 273.15) + 3.0)) < 1e-12
    assert np.isnan(po.isobar_temperature(p, t, 200.0))            # no extrapolation


def test_dewpoint_from_specific_humidity_chain():
    p, t, td = 900.0, 290.0, 283.0
    e = th.saturation_vapor_pressure(td)
    w = th.EPSILON * e / (p - e)
    q = w / (1.0 + w)
    # MetPy 1.4.1 goes through RH = w / w_s and e = RH e_s(T) = w (p - e_s(T)) / eps, which is not the exact inverse
    # of w = eps e / (p - e): the result sits 0.12 K below the dewpoint q was made from (the version drift noted in
    # the reference's env notebook); the closed form pins the chain
    got = th.dewpoint_from_specific_humidity(p, t, q)
    es = th.saturation_vapor_pressure(t)
    assert abs(got - th.dewpoint(w * (p - es) / th.EPSILON)) < 1e-10
    assert 0.05 < td - got < 0.2
    # saturated air: dewpoint equals temperature
    es = th.saturation_vapor_pressure(t)
    ws = th.EPSILON * es / (p - es)
    assert abs(th.dewpoint_from_specific_humidity(p, t, ws / (1 + ws)) - t) < 1e-10


def test_wind_shear_and_ship():
    h = np.array([100.0, 3000.0, 6000.0, 9000.0])
    u = np.array([2.0, 10.0, 20.0, 30.0])
    v = np.array([0.0, 0.0, 5.0, 5.0])
    ws = po.wind_shear(1.0, -1.0, u, v, h)
    assert ws['shear_u'] == 19.0 and ws['shear_v'] == 6.0 and ws['positive_shear']
    assert abs(ws['shear_magnitude'] - np.hypot(19.0, 6.0)) < 1e-12
    # SHIP: all thresholds satisfied -> plain product; a low freezing level scales it down
    ship = po.significant_hail_parameter(2000.0, 0.012, -7.0, 258.15, 20.0, 3000.0)
    assert abs(ship - 2000.0 * 12.0 * 7.0 * 15.0 * 20.0 / 42000000) < 1e-12
    assert abs(po.significant_hail_parameter(2000.0, 0.012, -7.0, 258.15, 20.0, 1200.0) - 0.5 * ship) < 1e-12
    assert np.isnan(po.significant_hail_parameter(2000.0, 0.012, -7.0, 258.15, 30.0, 3000.0))   # shear outside 7..27
    prox = po.storm_proxies({'shear_magnitude': [20.0], 'mixed_100_cape': [1200.0], 'mixed_50_cape': [1500.0], 'mu_cape': [2000.0],
                             'mixed_100_lifted_index': [-3.0], 'mixed_100_dci': [20.0], 'positive_shear': [True],
                             'mixed_50_cin': [-10.0], 'lapse_rate_700_500': [-7.0], 'mixed_100_cin': [-20.0],
                             'mu_mixing_ratio': [0.012], 'temp_500': [258.15], 'freezing_level': [3000.0]})
    assert all(bool(prox[k][0]) for k in prox if k != 'ship')
