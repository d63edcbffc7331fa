"""Kernel time of one rank's share of c4 / c5 (and c2) per moist mode: run_gpu_modes.py [c2|c4|c5] [modes...]"""
import sys, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
modes = sys.argv[2:] or ['exact', 'family']
nlev, ncol, dt, parcels = {'c2': (64, 1 << 20, torch.float64, ['surface']), 'c4': (128, 8192 * 8192 // 8, torch.float32, ['surface']),
                           'c5': (100, 24 * 2048 * 2048 // 8, torch.float32, ['surface', 'mixed_layer', 'most_unstable'])}[cfg]
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250721, dtype=dt)
out = {}
for parcel in parcels:
    for m in modes:
        ts = []
        for i in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = xa.cape_cin_columns(p, t, td, parcel=parcel, want=('cape', 'cin'), moist=m); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out[f'{cfg} {parcel} {m}'] = round(sorted(ts[1:])[2], 3)
print(json.dumps(out))
