"""family-mode c2-type call at ONE level count (the lowest n levels of the 128-level grid of run_gpu_levels.py), for counter runs"""
import sys
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev = int(sys.argv[1])
full = synth.columns_torch(128, 1 << 20, 'cuda', seed=20250719, dtype=torch.float64)
p, t, td = (v[:nlev].contiguous() for v in full)
for i in range(3):
    r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family')
torch.cuda.synchronize()
