import sys
sys.path.insert(0, '.')
import numpy as np
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co
np.set_printoptions(linewidth=220, precision=4, suppress=True)
def run(tag, p, t, td, **kw):
    got = xa.cape_cin_columns(p, t, td, moist='family', **kw); ref = co.cape_cin_grid(p, t, td, moist='family', **kw)
    bad = np.nonzero(got['el_index'] != ref['el_index'])[0]
    print(tag, 'bad', len(bad), bad[:8], 'waves', sorted(set((bad // 64).tolist()))[:10])
for nlev in (128,):
    p, t, td = synth.columns(nlev=nlev, ncol=1024, seed=20250720, dtype=np.float32, saturate_some=False)
    run('clean', p, t, td)
    t2 = t.copy(); td2 = td.copy(); t2[:, 70] = np.nan; td2[:, 70] = np.nan
    run('one all-NaN column 70', p, t2, td2)
    t2 = t.copy(); td2 = td.copy(); t2[0, 70] = np.nan
    run('NaN surface T col 70', p, t2, td2)
    t2 = t.copy(); td2 = td.copy(); t2[40:44, 70] = np.nan
    run('NaN T levels 40-43 col 70', p, t2, td2)
    td2 = td.copy(); td2[0, 70] = t[0, 70]
    run('saturated col 70', p, t, td2)
    p2 = p.copy(); p2[:, 70] *= 0.5
    run('half pressure col 70 (LCL high up)', p2, t, td)
    t2 = t.copy(); td2 = td.copy(); t2[:, 70] -= 95; td2[:, 70] -= 95
    run('cold col 70 (label outside)', p, t2, td2)
