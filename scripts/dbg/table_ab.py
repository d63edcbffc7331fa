"""Table mode (the reference's lookup tables) on c2 and the c4 / c5 share shapes: kernel ms + a hash of every output bit, to compare library builds
(XPARCEL_LIB=... python scripts/dbg/table_ab.py): the hashes of two builds must be equal."""
import sys, json, hashlib
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth, adiabat_tables
adiabat_tables.load_moist_adiabat_lookups(cache=False)
out = {}
for name, nlev, ncol, dt, parcels in (('c2', 64, 1 << 20, torch.float64, ['surface']), ('c4q', 128, 1 << 21, torch.float32, ['surface']),
                                      ('c5q', 100, 3 << 19, torch.float32, ['most_unstable', 'mixed_layer'])):
    p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250721, dtype=dt)
    for parcel in parcels:
        for _ in range(60): r = xa.cape_cin_columns(p, t, td, parcel=parcel, want=('cape', 'cin'), moist='table')   # warm clocks
        ts = []
        for i in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = xa.cape_cin_columns(p, t, td, parcel=parcel, want=('cape', 'cin'), moist='table'); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        full = xa.cape_cin_columns(p, t, td, parcel=parcel, moist='table')
        h = hashlib.sha256()
        for k in sorted(full):
            v = full[k]
            if torch.is_tensor(v): h.update(v.cpu().numpy().tobytes())
        for k in ('cape', 'cin'): h.update(r[k].cpu().numpy().tobytes())
        out[f'{name} {parcel}'] = [round(sorted(ts)[3], 4), h.hexdigest()[:12]]
print(json.dumps(out))
