"""c2 kernel in steady state (1000 warm calls, then the median of 60 event-timed calls): for same-box A/B of library builds"""
import sys
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
p, t, td = synth.columns_torch(64, 1024 * 1024, 'cuda', seed=20250719, dtype=torch.float64)
for i in range(1000): r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family')
torch.cuda.synchronize()
ev = []
for i in range(60):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family'); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print(round(ts[30], 4), round(ts[5], 4), round(ts[54], 4), 'cape_sum', float(torch.nan_to_num(r['cape']).sum()))
