"""Where persistent wavefronts start to pay (family mode): kernel time per grid size, ordinary launch vs persistent.
run_gpu_persist.py  -- spawns itself twice (XP_PERSIST_MIN_COLS is read once per process)."""
import sys, os, json, subprocess
sys.path.insert(0, '.')
SIZES = [(64, 1 << 20, 'f64'), (64, 2 << 20, 'f64'), (64, 4 << 20, 'f64'), (64, 8 << 20, 'f64'), (100, 2 << 20, 'f32'), (100, 4 << 20, 'f32')]
if len(sys.argv) > 1:
    import torch
    from xarray_parcel_amd import numpy_api as xa, synth
    out = {}
    for nlev, ncol, dt in SIZES:
        p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250721, dtype=torch.float64 if dt == 'f64' else torch.float32)
        for parcel in ('surface', 'most_unstable'):
            ts = []
            for i in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); xa.cape_cin_columns(p, t, td, parcel=parcel, want=('cape', 'cin'), moist='family'); e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            out['%d x %d %s %s' % (nlev, ncol, dt, parcel)] = round(sorted(ts[1:])[2], 3)
        del p, t, td
    print(sys.argv[1], json.dumps(out))
else:
    for name, v in (('ordinary', str(1 << 40)), ('persistent', '0')):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, XP_PERSIST_MIN_COLS=v), check=True)
