"""One combination of run_gpu_soak_profile.py in detail: prof_case.py seed nlev parcel dtype"""
import sys
sys.path.insert(0, '.')
import numpy as np
from oracle import c_oracle as co
from xarray_parcel_amd import numpy_api as xa, synth
seed, nlev, parcel, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], getattr(np, sys.argv[4])
kw = {'most_unstable': {'depth': 300}, 'mixed_layer': {'depth': 100}}.get(parcel, {})
xa.set_family_table(co.family_table())
p, t, td = synth.columns(nlev=nlev, ncol=60000, seed=seed * 11 + nlev, nan_fraction=0.06, dtype=dtype)
got = xa.cape_cin_columns(p, t, td, parcel=parcel, moist='family', want_profile=True, **kw)
ref = co.cape_cin_grid(p, t, td, parcel=parcel, moist='family', want_profile=True, **kw)
ex = xa.cape_cin_columns(p, t, td, parcel=parcel, moist='exact', want_profile=True, **kw)
cols = set()
for k in ('pressure', 'temperature', 'virtual_temperature', 'environment_temperature', 'environment_virtual_temperature', 'environment_dewpoint'):
    a, b = np.asarray(got['profile'][k], dtype=np.float64), np.asarray(ref['profile'][k], dtype=np.float64)
    m = min(a.shape[0], b.shape[0])
    d = np.abs(a[:m] - b[:m]); d[np.isnan(d)] = 0
    j, c = np.unravel_index(np.argmax(d), d.shape)
    print(k, 'max dev', d[j, c], 'row', j, 'col', c, 'gpu', a[j, c], 'oracle', b[j, c], 'exact-mode gpu', float(np.asarray(ex['profile'][k], dtype=np.float64)[j, c]))
    if d[j, c] > 1e-9: cols.add(int(c))
for c in sorted(cols):
    print('column', c, {k: (float(got[k][c]), float(ref[k][c])) for k in ('lcl_pressure', 'lcl_temperature', 'cape', 'cin')}, 'status', int(got['status'][c]), int(ref['status'][c]))
    a = np.asarray(got['profile']['pressure'])[:, c]; b = np.asarray(ref['profile']['pressure'])[:, c]
    j = int(np.nanargmin(np.abs(a - float(got['lcl_pressure'][c]))))
    print('  rows around the LCL: gpu p', a[max(0, j - 2):j + 3], 'oracle p', b[max(0, j - 2):j + 3])
    for k in ('temperature', 'virtual_temperature', 'environment_dewpoint'):
        print('   ', k, np.asarray(got['profile'][k])[max(0, j - 2):j + 3, c], np.asarray(ref['profile'][k])[max(0, j - 2):j + 3, c])
