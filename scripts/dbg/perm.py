"""permutation equivariance of CAPE / CIN on the c2 grid (tests/test_gpu_parity.py::test_full_size_properties_config2), which columns differ"""
import sys
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev, ncol = 64, 1024 * 1024
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250719, dtype=torch.float64)
for moist in ('family', 'exact'):
    a = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist=moist)
    perm = torch.randperm(ncol, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
    q = xa.cape_cin_columns(p[:, perm].contiguous(), t[:, perm].contiguous(), td[:, perm].contiguous(), want=('cape', 'cin'), moist=moist)
    for k in ('cape', 'cin'):
        d = (q[k] != a[k][perm]).nonzero().flatten()
        print(moist, k, 'differing columns', int(d.numel()), 'max abs diff', float((q[k] - a[k][perm]).abs().max()), d[:5].tolist())
