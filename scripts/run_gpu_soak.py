"""Parity soak: many seeds x level counts x parcels x option sets x dtypes against the C oracle, with the tie
classification of tests/test_gpu_parity.py.  Prints one line per combination and a summary; exits non-zero on a mismatch."""
import sys, itertools, time
sys.path.insert(0, '.')
import numpy as np
from oracle import c_oracle as co
from tests import test_gpu_parity as tp
from xarray_parcel_amd import numpy_api as xa, synth
tp.xa = xa
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
seeds = range(100, 100 + (int(sys.argv[2]) if len(sys.argv) > 2 else 4))
variant = sys.argv[3] if len(sys.argv) > 3 else 'exact'        # exact | family | table | specific (q input, fused conversion) | fused (xp_cape_cin_multi, one pass for two parcels)
fused = variant == 'fused'
if fused:
    variant = 'family'
if variant == 'family':
    xa.set_family_table(co.family_table())                     # both sides interpolate the oracle's table
if variant == 'table':
    from oracle import tables as otb
    from xarray_parcel_amd import adiabat_tables
    tab = otb.get_tables()
    co.set_tables(tab)
    adiabat_tables.set_tables(tab.index, tab.adiabats)         # both sides look up the same arrays
from oracle import thermo as th
bad = 0; n = 0; t0 = time.time()
for seed, nlev, parcel, mode, dtype in itertools.product(seeds, (9, 33, 64, 100), ('surface', 'most_unstable', 'mixed_layer'),
                                                         range(len(tp.MODES)), (np.float64, np.float32)):
    kw = tp.MODES[mode]
    p, t, td = synth.columns(nlev=nlev, ncol=ncol, seed=seed * 7 + nlev, nan_fraction=0.08, dtype=dtype)
    if variant == 'specific':
        e = th.saturation_vapor_pressure(td.astype(np.float64)); w = th.EPSILON * e / (p - e)
        q = (w / (1.0 + w)).astype(dtype)
        with np.errstate(all='ignore'):
            td_ref = th.dewpoint_from_specific_humidity(p.astype(np.float64), t.astype(np.float64), q.astype(np.float64))
        got = xa.cape_cin_columns(p, t, q, parcel=parcel, humidity='specific', **kw)
        ref = co.cape_cin_grid(p.astype(np.float64), t.astype(np.float64), td_ref, parcel=parcel, moist='rk4', **kw)
    else:
        got = xa.cape_cin_columns(p, t, td, parcel=parcel, moist=variant, **kw)
        ref = co.cape_cin_grid(p, t, td, parcel=parcel, moist='rk4' if variant == 'exact' else variant, **kw)
    n += 1
    if fused and parcel != 'surface':             # the pair (this parcel, surface) in ONE pass against the separate calls, bit for bit
        try:
            pair = xa.cape_cin_multi(p, t, td, [(parcel, None), ('surface', None)], moist='family', fused=True, **kw)
            sfc = xa.cape_cin_columns(p, t, td, parcel='surface', moist='family', **kw)
            for g_, r_ in ((pair[0], got), (pair[1], sfc)):
                for k in r_:
                    assert np.array_equal(np.asarray(g_[k]), np.asarray(r_[k]), equal_nan=True), ('fused pass differs', k)
        except AssertionError as e:
            bad += 1
            print('MISMATCH (fused)', seed, nlev, parcel, mode, dtype.__name__, str(e)[:300], flush=True)
    try:
        if mode == 0 and variant != 'specific':      # the CAPE / CIN-only (LEAN) instantiation: bit-identical to the all-outputs kernel
            lean = xa.cape_cin_columns(p, t, td, parcel=parcel, moist=variant, want=('cape', 'cin', 'lfc_pressure', 'el_pressure'))
            for k in lean:
                assert np.array_equal(np.asarray(lean[k]), np.asarray(got[k]), equal_nan=True), ('lean kernel differs', k)
        tp._compare(got, ref, dtype, 1e-6)
    except AssertionError as e:
        bad += 1
        print('MISMATCH', seed, nlev, parcel, mode, dtype.__name__, str(e)[:300], flush=True)
    if n % 24 == 0:
        print('progress', n, 'combos', round(time.time() - t0), 's', flush=True)
print('SOAK', n, 'combinations x', ncol, 'columns:', bad, 'mismatching combinations')
sys.exit(1 if bad else 0)
