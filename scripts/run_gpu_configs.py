"""Evidence run for BASELINE configs c3, c4 (one rank's share), c5 (one rank's share) in both exact moist modes: kernel time
(HIP events, median of 3 after a warm-up), roofline fraction on the algorithmic bytes of SURVEY.md 8(d), and a strided
sample against the C oracle.  Writes one JSON object per config to stdout (committed under profiles/ by hand)."""
import sys, json
sys.path.insert(0, '.')
import numpy as np, torch
from xarray_parcel_amd import numpy_api as xa, synth
from oracle import c_oracle as co

def timed(fn, n=4):
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        if i < n - 1: del r
    return sorted(ts[1:])[len(ts[1:]) // 2], r

def sample_check(r, p, t, td, step, parcel, omode, profile=False):
    idx = torch.arange(0, p.shape[1], step, device='cuda')
    ref = co.cape_cin_grid(p[:, idx].cpu().numpy(), t[:, idx].cpu().numpy(), td[:, idx].cpu().numpy(), parcel=parcel, moist=omode, want_profile=profile)
    out = {'columns_checked': int(idx.numel())}
    keys = ('lfc_index', 'el_index') + (('parcel_index',) if parcel != 'surface' else ())
    out['indices_identical'] = bool(all(np.array_equal(r[k][idx].cpu().numpy(), ref[k]) for k in keys))
    for k in ('cape', 'cin'):
        out[k + '_maxdiff'] = float(np.max(np.abs(r[k][idx].cpu().numpy().astype(np.float64) - ref[k])))
    if profile:
        pr = r['profile']['temperature'][:, idx].cpu().numpy().astype(np.float64)
        out['profile_T_maxdiff'] = float(np.nanmax(np.abs(pr - ref['profile']['temperature'])))
    return out

which = sys.argv[1:] or ['c3', 'c4', 'c5']
res = {}
if 'c3' in which:
    nlev, ncol = 128, 4096 * 4096
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250720, dtype=torch.float32)
    alg = (3 * nlev * 4 + 6 * (nlev + 1) * 4 + 13 * 4) * ncol
    for m, om in (('family', 'family'), ('exact', 'rk4')):
        ms, r = timed(lambda: xa.cape_cin_columns(p, t, td, want_profile=True, moist=m))
        res[f'c3 {m}'] = {'config': 'c3: 128 x 4096 x 4096 fp32, full profile + LCL/LFC/EL + CAPE/CIN', 'kernel_ms': ms, 'columns_per_s': ncol / ms * 1e3,
                          'algorithmic_GBs': alg / ms / 1e6, 'frac_of_8TBs': alg / ms / 1e6 / 8000, 'sample': sample_check(r, p, t, td, 4099, 'surface', om, True)}
        del r
    del p, t, td
if 'c4' in which:
    nlev, ncol = 128, 8192 * 8192 // 8
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250721, dtype=torch.float32)
    alg = (3 * nlev * 4 + 2 * 4) * ncol
    for m, om in (('family', 'family'), ('exact', 'rk4')):
        ms, r = timed(lambda: xa.cape_cin_columns(p, t, td, want=('cape', 'cin', 'lfc_index', 'el_index'), moist=m))
        res[f'c4 share {m}'] = {'config': 'c4, one of 8 ranks: 128 x 8.4M columns fp32, surface-based CAPE/CIN', 'kernel_ms': ms, 'columns_per_s': ncol / ms * 1e3,
                                'algorithmic_GBs': alg / ms / 1e6, 'frac_of_8TBs': alg / ms / 1e6 / 8000, 'gather_payload_MB_per_rank': 2 * 4 * ncol / 1e6,
                                'sample': sample_check(r, p, t, td, 2003, 'surface', om)}
    del p, t, td
if 'c5' in which:
    nlev, ncol = 100, 24 * 2048 * 2048 // 8
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250722, dtype=torch.float32)
    alg = (3 * nlev * 4 + 2 * 4) * ncol
    for parcel in ('most_unstable', 'mixed_layer', 'surface'):
        for m, om in (('family', 'family'), ('exact', 'rk4')):
            ms, r = timed(lambda: xa.cape_cin_columns(p, t, td, parcel=parcel, want=('cape', 'cin', 'lfc_index', 'el_index', 'parcel_index'), moist=m))
            res[f'c5 share {parcel} {m}'] = {'config': f'c5, one of 8 ranks: 100 x 12.6M columns fp32, {parcel} CAPE/CIN', 'kernel_ms': ms, 'columns_per_s': ncol / ms * 1e3,
                                             'algorithmic_GBs': alg / ms / 1e6, 'frac_of_8TBs': alg / ms / 1e6 / 8000,
                                             'sample': sample_check(r, p, t, td, 3001, parcel, om)}
for k, v in res.items():
    print(json.dumps({k: v}))
