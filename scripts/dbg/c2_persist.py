import sys, os, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev = 64
for ncol in (1 << 20, 1 << 22):
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250719, dtype=torch.float64)
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family'); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    print(json.dumps({'persist_min': os.environ.get('XP_PERSIST_MIN_COLS'), 'ncol': ncol, 'ms_median': ts[len(ts) // 2], 'ms_min': ts[0], 'ms_per_Mi': ts[len(ts) // 2] / (ncol / (1 << 20))}))
