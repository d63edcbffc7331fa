"""The product bundle (conv_properties, pf.py:1951) on a device-resident synthetic grid: total time, and the three parcel
passes with the full six-array profile against the three arrays lifted_index reads.  run_gpu_bundle.py [nlev] [ncol]"""
import sys, time, json
sys.path.insert(0, '.')
import numpy as np
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250722, dtype=torch.float64)
e = 6.112 * torch.exp(17.67 * (td - 273.15) / (td - 29.65)); w = 0.6219569100577033 * e / (p - e); q = w / (1 + w)
z = 44330.8 * (1.0 - (p / 1013.25) ** 0.190263)                     # any monotone height will do here
g = torch.Generator(device='cuda').manual_seed(3)
nw = 12
wh = torch.linspace(50.0, 9000.0, nw, device='cuda', dtype=torch.float64)[:, None] + 40 * torch.rand((1, ncol), device='cuda', dtype=torch.float64, generator=g)
d = {'pressure': p, 'temperature': t, 'specific_humidity': q, 'height_asl': z, 'wind_u': 5.0 + wh * 2.5e-3, 'wind_v': -2.0 + wh * 1e-3,
     'wind_height_above_surface': wh, 'surface_wind_u': torch.zeros(ncol, device='cuda', dtype=torch.float64) + 2.0,
     'surface_wind_v': torch.zeros(ncol, device='cuda', dtype=torch.float64)}


def timed(f, n=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[len(ts) // 2] * 1e3, 2)


def parcels(want):
    xa.cape_cin_columns(p, t, td, parcel='most_unstable', depth=250, want_profile=want)
    xa.cape_cin_columns(p, t, td, parcel='mixed_layer', depth=100, want_profile=want)
    xa.cape_cin_columns(p, t, td, parcel='mixed_layer', depth=50, want_profile=want)


out = {'grid': [nlev, ncol], 'conv_properties_ms': timed(lambda: xa.conv_properties(d)),
       'conv_properties_family_ms': timed(lambda: xa.conv_properties(d, moist='family')),
       'conv_properties_composed_ms': timed(lambda: xa.conv_properties_composed(d)),
       'three_parcel_passes_all_six_arrays_ms': timed(lambda: parcels(True)),
       'three_parcel_passes_lifted_index_arrays_ms': timed(lambda: parcels(xa.LIFTED_INDEX_VARS)),
       'three_parcel_passes_no_profile_ms': timed(lambda: parcels(False))}
print(json.dumps(out))
