"""One rank's share of BASELINE config c4 (128-level x 8192 x 8192 fp32 over 8 GPUs = 8.4 M columns per GPU): kernel time for
surface-based CAPE/CIN only + the size of the per-step gather payload."""
import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co
nlev, ncol = 128, 8192 * 8192 // 8
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250721, dtype=torch.float32)
ts = []
for i in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin', 'lfc_index', 'el_index')); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ms = sorted(ts[1:])[2]
idx = torch.arange(0, ncol, 2003, device='cuda')
ref = co.cape_cin_grid(p[:, idx].cpu().numpy(), t[:, idx].cpu().numpy(), td[:, idx].cpu().numpy(), moist='rk4')
ok = all(np.array_equal(r[k][idx].cpu().numpy(), ref[k]) for k in ('lfc_index', 'el_index'))
alg = (3 * nlev * 4 + 2 * 4) * ncol
print(json.dumps({'config': 'c4 share: 128 x 8.4M fp32 CAPE/CIN', 'kernel_ms': ms, 'columns_per_s': ncol / ms * 1e3,
                  'algorithmic_GBs': alg / ms / 1e6, 'frac_of_8TBs': alg / ms / 1e6 / 8000, 'indices_match_sample': bool(ok),
                  'cape_maxdiff': float(np.max(np.abs(r['cape'][idx].cpu().numpy().astype(np.float64) - ref['cape']))),
                  'gather_payload_MB_per_rank': 2 * 4 * ncol / 1e6}))
