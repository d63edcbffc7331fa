"""conv_properties (family mode): wall time against the GPU time between two events, on the bundle grid and on a tiny grid (host cost per call)"""
import sys, time, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
def inputs(nlev, ncol):
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250722, dtype=torch.float64)
    e = 6.112 * torch.exp(17.67 * (td - 273.15) / (td - 29.65)); w = 0.6219569100577033 * e / (p - e); q = w / (1 + w)
    z = 44330.8 * (1.0 - (p / 1013.25) ** 0.190263)
    nw = 12
    wh = torch.linspace(50.0, 9000.0, nw, device='cuda', dtype=torch.float64)[:, None] + torch.zeros((1, ncol), device='cuda', dtype=torch.float64)
    return {'pressure': p, 'temperature': t, 'specific_humidity': q, 'height_asl': z, 'wind_u': 5.0 + wh * 2.5e-3, 'wind_v': -2.0 + wh * 1e-3,
            'wind_height_above_surface': wh, 'surface_wind_u': torch.zeros(ncol, device='cuda', dtype=torch.float64) + 2.0,
            'surface_wind_v': torch.zeros(ncol, device='cuda', dtype=torch.float64)}
out = {}
for name, ncol in (('1Mi', 1 << 20), ('1Ki', 1 << 10)):
    d = inputs(64, ncol)
    for _ in range(20): xa.conv_properties(d, moist='family')
    torch.cuda.synchronize()
    ws, es = [], []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); xa.conv_properties(d, moist='family'); e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ws.append((t1 - t0) * 1e3); es.append(e0.elapsed_time(e1))
    out[name] = {'host_call_ms': round(sorted(ws)[4], 3), 'gpu_between_events_ms': round(sorted(es)[4], 3)}
print(json.dumps(out))
