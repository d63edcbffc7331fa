"""Per-wavefront lifetimes of the persistent c2 kernel (diagnostic build -DXP_WAVE_TIMES: XPARCEL_LIB=.../lib_wt.so).
Prints how long the wavefronts live relative to the kernel, per-workgroup and grid-wide spread of their end times."""
import sys, os, json
sys.path.insert(0, '.')
import numpy as np, torch
from xarray_parcel_amd import numpy_api as xa, synth
smooth = len(sys.argv) > 1 and sys.argv[1] == 'smooth'
p, t, td = synth.columns_torch(64, 1 << 20, 'cuda', seed=20250719, dtype=torch.float64)
for i in range(1200): r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family')     # warm clocks (file not yet asked for)
torch.cuda.synchronize()
ts = []
for i in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family'); e1.record()
    ts.append((e0, e1))
torch.cuda.synchronize()
call_ms = sorted(a.elapsed_time(b) for a, b in ts)[20]
os.environ['XP_WAVE_TIMES_FILE'] = '/tmp/wt.bin'
r = xa.cape_cin_columns(p, t, td, want=('cape', 'cin'), moist='family')
torch.cuda.synchronize()
d = np.fromfile('/tmp/wt.bin', dtype=np.uint64).reshape(-1, 5)[:4096].astype(np.int64)
t0, t1, t2, n, hw = d.T
base = t0.min()
tick = 10.0   # ns per 100 MHz tick
out = {'call_ms_median_of_40': call_ms, 'kernel_span_us': (t2.max() - base) * tick / 1e3, 'start_spread_us': (t0.max() - base) * tick / 1e3,
       'staged_after_us_mean': ((t1 - base).mean()) * tick / 1e3, 'staged_after_us_max': (t1 - base).max() * tick / 1e3,
       'end_us_mean': (t2 - base).mean() * tick / 1e3, 'end_us_min': (t2 - base).min() * tick / 1e3, 'end_us_p10': float(np.percentile(t2 - base, 10)) * tick / 1e3,
       'end_us_p50': float(np.percentile(t2 - base, 50)) * tick / 1e3, 'end_us_p90': float(np.percentile(t2 - base, 90)) * tick / 1e3,
       'tiles_hist': np.bincount(n, minlength=8)[:8].tolist()}
wg = (t2 - base).reshape(256, 16)
out['workgroup_end_us: mean of max'] = wg.max(1).mean() * tick / 1e3
out['workgroup_end_us: min / max of max'] = [wg.max(1).min() * tick / 1e3, wg.max(1).max() * tick / 1e3]
out['workgroup_end_us: mean of (max - mean)'] = (wg.max(1) - wg.mean(1)).mean() * tick / 1e3
out['idle_fraction'] = 1.0 - (t2 - t1).sum() / ((t2.max() - base) * 4096.0)
print(json.dumps(out, indent=1))
