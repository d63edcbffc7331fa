"""One rank's share of BASELINE config c5 (100 levels x 12.6 M columns fp32): most-unstable + mixed-layer CAPE / CIN as two
separate xp_cape_cin calls and as one fused xp_cape_cin_multi call (family mode): HIP-event times, bitwise comparison of
the two, strided sample against the oracle.  Usage: run_gpu_multi.py [ncol_divisor]"""
import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co
div = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nlev, ncol = 100, 24 * 2048 * 2048 // 8 // div
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250722, dtype=torch.float32)
want = ('cape', 'cin')
parcels = [('most_unstable', 300.0), ('mixed_layer', 100.0)]

def timed(fn, n=5):
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts[1:])[len(ts[1:]) // 2], r

out = {'ncol': ncol, 'nlev': nlev}
sep = {}
for name, depth in parcels:
    ms, r = timed(lambda: xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, want=want, moist='family'))
    sep[name] = r
    out['separate_' + name + '_ms'] = ms
out['separate_sum_ms'] = sum(out['separate_' + n + '_ms'] for n, _ in parcels)
ms, fused = timed(lambda: xa.cape_cin_multi(p, t, td, parcels, want=want, moist='family', fused=True))
out['fused_ms'] = ms
alg = (3 * nlev * 4 + 2 * 4) * ncol
out['fused_frac_of_8TBs_single_parcel_bytes'] = alg / ms / 1e6 / 8000
for (name, depth), g in zip(parcels, fused):
    for k in want:
        a, b = g[k], sep[name][k]
        out[f'bitwise_{name}_{k}'] = bool(torch.equal(a, b) or (torch.equal(torch.isnan(a), torch.isnan(b)) and bool((a[~torch.isnan(a)] == b[~torch.isnan(b)]).all())))
idx = torch.arange(0, ncol, 3001, device='cuda')
full = xa.cape_cin_multi(p[:, idx].contiguous(), t[:, idx].contiguous(), td[:, idx].contiguous(), parcels, moist='family', fused=True)
for (name, depth), g, f in zip(parcels, fused, full):
    ref = co.cape_cin_grid(p[:, idx].cpu().numpy(), t[:, idx].cpu().numpy(), td[:, idx].cpu().numpy(), parcel=name, depth=depth, moist='family')
    out[f'oracle_{name}'] = {'columns': int(idx.numel()),
                             'indices_identical': bool(all(np.array_equal(f[k].cpu().numpy(), ref[k]) for k in ('lfc_index', 'el_index', 'parcel_index'))),
                             'cape_maxdiff': float(np.max(np.abs(g['cape'][idx].cpu().numpy().astype(np.float64) - ref['cape']))),
                             'cin_maxdiff': float(np.max(np.abs(g['cin'][idx].cpu().numpy().astype(np.float64) - ref['cin'])))}
print(json.dumps(out))
