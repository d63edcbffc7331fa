"""family-mode profile temperature of a seeded fp64 grid -> npy (to compare library builds); also the distance to the C oracle on a sample"""
import sys
sys.path.insert(0, '.')
import numpy as np
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co
out = sys.argv[1]
xa.set_family_table(co.family_table())
res = {}
for parcel, kw in (('surface', {}), ('most_unstable', {'depth': 300}), ('mixed_layer', {'depth': 100})):
    p, t, td = synth.columns(nlev=100, ncol=60000, seed=41, nan_fraction=0.05, dtype=np.float64)
    r = xa.cape_cin_columns(p, t, td, parcel=parcel, moist='family', want_profile=True, **kw)
    res[parcel] = np.asarray(r['profile']['temperature'])
    ref = co.cape_cin_grid(p[:, :4000], t[:, :4000], td[:, :4000], parcel=parcel, moist='family', want_profile=True, **kw)
    a, b = res[parcel][:, :4000], ref['profile']['temperature']
    n = min(a.shape[0], b.shape[0])
    ok = ~np.isnan(b[:n])
    print(parcel, 'nan pattern equal', bool(np.array_equal(np.isnan(a[:n]), np.isnan(b[:n]))), 'max |T - oracle|', float(np.max(np.abs(a[:n][ok] - b[:n][ok]))))
np.savez(out, **res)
