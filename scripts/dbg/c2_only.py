"""c2: surface CAPE/CIN-only family kernel on 64 x 1 Mi fp64 columns (for the PMC script)"""
import sys
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
p, t, td = synth.columns_torch(64, 1024 * 1024, 'cuda', seed=20250719, dtype=torch.float64)
for i in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family'); e1.record(); torch.cuda.synchronize()
print(e0.elapsed_time(e1))
