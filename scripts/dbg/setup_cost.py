"""Surface CAPE/CIN-only family kernel on 1 Mi fp64 columns with 2 and with 64 levels: what the per-column set-up and finish
cost in instructions (run under scripts/run_gpu_pmc_any.sh)."""
import sys
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
ncol = 1024 * 1024
for nlev in (2, 64):
    p, t, td = synth.columns_torch(64, ncol, 'cuda', seed=20250719, dtype=torch.float64)
    p, t, td = p[:nlev].contiguous(), t[:nlev].contiguous(), td[:nlev].contiguous()
    for i in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family'); e1.record(); torch.cuda.synchronize()
    print(nlev, e0.elapsed_time(e1))
