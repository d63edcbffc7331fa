import sys, os, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev = 64
out = {}
for ncol in (1 << 18, 1 << 19, 3 << 18, 1 << 20, 5 << 18, 3 << 19, 1 << 21, 1 << 22):
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250719, dtype=torch.float64)
    ts = []
    for i in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family'); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[3:])
    out[ncol >> 18] = round(ts[len(ts) // 2], 4)
print(json.dumps(out))
