"""c3-shaped launches (128 levels fp32, full profile + scalars) on a quarter grid, both ODE modes: for the PMC script."""
import sys
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev, ncol = 128, 4 * 1024 * 1024
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250720, dtype=torch.float32)
for moist in ('exact', 'family'):
    for i in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = xa.cape_cin_columns(p, t, td, want_profile=True, moist=moist); e1.record(); torch.cuda.synchronize()
        del r
    print(moist, e0.elapsed_time(e1))
