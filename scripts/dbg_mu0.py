import sys
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
p, t, td = synth.columns_torch(64, 1 << 20, 'cuda', seed=20250719, dtype=torch.float64)
for parcel, kw in (('surface', {}), ('most_unstable', {'depth': 0.0}), ('surface', {})):
    ts = []
    for i in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = xa.cape_cin_columns(p, t, td, parcel=parcel, want=('cape', 'cin'), moist='family', **kw); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(parcel, kw, sorted(ts)[3], float(r['cape'].sum()))
