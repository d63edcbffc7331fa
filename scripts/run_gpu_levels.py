"""Kernel time against the number of levels at a fixed column count (fixed per-column cost vs per-level cost)."""
import sys, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
ncol = 1 << 20
levels = [int(v) for v in sys.argv[1:]] or [16, 32, 64, 128]
out = {}
full = synth.columns_torch(128, ncol, 'cuda', seed=20250719, dtype=torch.float64)
for nlev in levels:
    # the LOWEST nlev levels of one 128-level grid: the same parcels and LCLs whatever nlev is
    p, t, td = (v[:nlev].contiguous() for v in full)
    for m in ('family', 'exact'):
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist=m); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out[f'{m} {nlev}'] = round(sorted(ts)[3], 3)
print(json.dumps(out))
