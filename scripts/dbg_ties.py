import sys
sys.path.insert(0, '.')
import numpy as np
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co
for (nlev, ncol, seed, dt, nf) in ((128, 20000, 20250721, np.float32, 0.02), (48, 12000, 7, np.float64, 0.08)):
    p, t, td = synth.columns(nlev=nlev, ncol=ncol, seed=seed, nan_fraction=nf, dtype=dt)
    got = xa.cape_cin_columns(p, t, td)
    ref = co.cape_cin_grid(p, t, td, moist='rk4')
    sat = ref['lcl_pressure'] == np.asarray(got['parcel_pressure'], dtype=np.float64)
    gi, ri = np.asarray(got['lfc_index']), ref['lfc_index']
    tie = sat & (gi != ri)
    print(nlev, 'saturated', int(sat.sum()), 'ties', int(tie.sum()), 'got=-2', int((tie & (gi == -2)).sum()), 'ref=-2', int((tie & (ri == -2)).sum()),
          'maxdiff lfc p on ties', float(np.nanmax(np.abs(np.asarray(got['lfc_pressure'])[tie] - ref['lfc_pressure'][tie]))) if tie.any() else None)
