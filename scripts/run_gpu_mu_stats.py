"""Most-unstable parcel: where the selected level sits in the synthetic c5 columns, and how far the lanes of a wavefront
are apart (the level loop starts at each lane's own parcel level)."""
import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev, ncol = 100, 1 << 20
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250722, dtype=torch.float32)
r = xa.cape_cin_columns(p, t, td, parcel='most_unstable', want=('parcel_index',))
idx = r['parcel_index'].cpu().numpy()
h = np.bincount(np.clip(idx, 0, 60), minlength=61)
w = idx.reshape(-1, 64)
print(json.dumps({'frac_surface': float((idx == 0).mean()), 'mean_index': float(idx.mean()), 'p99': int(np.percentile(idx, 99)),
                  'max': int(idx.max()), 'wave_max_mean': float(w.max(1).mean()), 'wave_distinct_mean': float(np.mean([len(set(x)) for x in w[:4000]])),
                  'hist_0_40': h[:41].tolist()}))
