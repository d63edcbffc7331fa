"""BASELINE config c3: synthetic 128-level x 4096 x 4096 fp32 soundings, full parcel profile + LCL/LFC/EL + CAPE/CIN.
Times the kernel (HIP events) and checks size-independent properties + a strided sample against the oracle."""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co
nlev, ny, nx = 128, 4096, 4096
ncol = ny * nx
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250720, dtype=torch.float32)
torch.cuda.synchronize()
times = []
for i in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = xa.cape_cin_columns(p, t, td, want_profile=True); e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
    if i < 3: del r
ms = sorted(times[1:])[len(times[1:]) // 2]
alg = (3 * nlev * 4 + 6 * (nlev + 1) * 4 + 13 * 4) * ncol
idx = torch.arange(0, ncol, 4099, device='cuda')
ref = co.cape_cin_grid(p[:, idx].cpu().numpy(), t[:, idx].cpu().numpy(), td[:, idx].cpu().numpy(), moist='rk4', want_profile=True)
ok = {}
for k in ('lfc_index', 'el_index'):
    ok[k] = bool(np.array_equal(r[k][idx].cpu().numpy(), ref[k]))
for k in ('cape', 'cin'):
    ok[k + '_maxdiff'] = float(np.max(np.abs(r[k][idx].cpu().numpy().astype(np.float64) - ref[k])))
pr = r['profile']['temperature'][:, idx].cpu().numpy().astype(np.float64)
ok['profile_T_maxdiff'] = float(np.nanmax(np.abs(pr - ref['profile']['temperature'])))
print(json.dumps({'config': 'c3 128x4096x4096 fp32 full profile', 'kernel_ms': ms, 'columns_per_s': ncol / ms * 1e3,
                  'algorithmic_GBs': alg / ms / 1e6, 'frac_of_8TBs': alg / ms / 1e6 / 8000, 'sample_check': ok}))
