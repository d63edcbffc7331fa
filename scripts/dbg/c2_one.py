import sys, os, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev, ncol = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250719, dtype=torch.float64)
for i in range(6):
    r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family')
torch.cuda.synchronize()
