import sys
sys.path.insert(0, '.')
import numpy as np
from xarray_parcel_amd import numpy_api as xa, synth
np.set_printoptions(linewidth=200, precision=6)
p, t, td = synth.columns(nlev=48, ncol=12000, seed=11, nan_fraction=0.08, dtype=np.float64)
parcels = [('most_unstable', 300.0), ('mixed_layer', 100.0)]
got = xa.cape_cin_multi(p, t, td, parcels, moist='family')
for (name, depth), g in zip(parcels, got):
    ref = xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, moist='family')
    bad = np.zeros(12000, bool)
    for k in ref:
        x, y = np.asarray(g[k]), np.asarray(ref[k])
        d = ~((x == y) | ((x != x) & (y != y)))
        if d.any(): print(name, k, 'differs in', int(d.sum()))
        bad |= d
    for c in np.nonzero(bad)[0][:4]:
        print('column', c)
        for k in ref: print('   ', k, np.asarray(g[k])[c], np.asarray(ref[k])[c])
        print('   p', p[:22, c]); print('   t', t[:22, c]); print('   td', td[:22, c])
