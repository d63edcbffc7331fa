"""NaN PRESSURE inside a column: what the kernel does vs the oracle's literal insert_level (pf.py:962-966).  Prints per case."""
import sys
sys.path.insert(0, '.')
import numpy as np
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co
np.set_printoptions(linewidth=200, precision=6)
p, t, td = synth.columns(nlev=30, ncol=64, seed=9, dtype=np.float64)
base = co.cape_cin_grid(p, t, td, moist='rk4', want_profile=True)
# level index of the first level above the LCL per column
first_above = np.array([int(np.argmax(p[:, c] < base['lcl_pressure'][c])) for c in range(64)])
cases = {}
q = p.copy()
for c in range(64):
    k = first_above[c]
    kind = c % 4
    if kind == 0 and k >= 2: q[k - 1, c] = np.nan           # NaN pressure just below the LCL
    elif kind == 1 and k >= 3: q[1, c] = np.nan             # NaN pressure well below the LCL
    elif kind == 2: q[min(k + 2, 29), c] = np.nan           # NaN pressure above the LCL
    # kind 3: untouched
got = xa.cape_cin_columns(q, t, td, want_profile=True)
ref = co.cape_cin_grid(q, t, td, moist='rk4', want_profile=True)
for kind, name in enumerate(('just below LCL', 'well below LCL', 'above LCL', 'untouched')):
    cols = [c for c in range(64) if c % 4 == kind]
    dc = np.abs(got['cape'][cols] - ref['cape'][cols]); dn = np.abs(got['cin'][cols] - ref['cin'][cols])
    print(name, 'status', sorted(set(got['status'][cols].tolist())), 'max|dCAPE|', dc.max(), 'max|dCIN|', dn.max(),
          'lfc_idx equal', np.array_equal(got['lfc_index'][cols], ref['lfc_index'][cols]),
          'el_idx equal', np.array_equal(got['el_index'][cols], ref['el_index'][cols]))
c = 0
print('column 0 (NaN p just below LCL): first_above', first_above[c], 'lcl', base['lcl_pressure'][c])
for k in ('pressure', 'temperature', 'environment_temperature'):
    print(k, 'got', got['profile'][k][:12, c]); print(k, 'ref', ref['profile'][k][:12, c])
print('cape got/ref', got['cape'][c], ref['cape'][c], 'cin', got['cin'][c], ref['cin'][c], 'lfc p', got['lfc_pressure'][c], ref['lfc_pressure'][c])
c = 1
print('column 1 (NaN p well below LCL):')
for k in ('pressure', 'temperature'):
    print(k, 'got', got['profile'][k][:12, c]); print(k, 'ref', ref['profile'][k][:12, c])
print('cape got/ref', got['cape'][c], ref['cape'][c], 'cin', got['cin'][c], ref['cin'][c])
