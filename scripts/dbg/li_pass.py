"""The bundle's three parcel passes (64 x 1 Mi fp64) with and without the in-pass lifted index, per moist mode: ms (median of 5 after a warm-up)"""
import sys, time, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
p, t, td = synth.columns_torch(64, 1 << 20, 'cuda', seed=20250722, dtype=torch.float64)
def passes(moist, li):
    kw = dict(lifted_index_at=500.0) if li else {}
    xa.cape_cin_columns(p, t, td, parcel='most_unstable', depth=250, want=('cape', 'cin'), moist=moist, **kw)
    xa.cape_cin_columns(p, t, td, parcel='mixed_layer', depth=100, want=('cape', 'cin'), moist=moist, **kw)
    xa.cape_cin_columns(p, t, td, parcel='mixed_layer', depth=50, want=('cape', 'cin'), moist=moist, **kw)
out = {}
for moist in ('family', 'exact'):
    for li in (False, True):
        for _ in range(30): passes(moist, li)
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            t0 = time.perf_counter(); passes(moist, li); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out[f'{moist} li={li}'] = round(sorted(ts)[2] * 1e3, 3)
print(json.dumps(out))
