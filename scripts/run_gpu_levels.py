"""Kernel time against the number of levels at a fixed column count (fixed per-column cost vs per-level cost)."""
import sys, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
ncol = 1 << 21
out = {}
for nlev in (16, 32, 64, 128):
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250719, dtype=torch.float64)
    for m in ('family', 'exact'):
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist=m); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out[f'{m} {nlev}'] = round(sorted(ts)[3], 3)
    del p, t, td
print(json.dumps(out))
