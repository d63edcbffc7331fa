"""Device-resident 64 x 1 Mi fp64 grid: ms per call of the array API's column functions (median of 5 after 3 warm-ups), with the bytes each
must move at least -- to spot a function that is far from its traffic."""
import sys, time, json
sys.path.insert(0, '.')
import torch
from xarray_parcel_amd import numpy_api as xa, synth
nlev, ncol = 64, 1 << 20
p, t, td = synth.columns_torch(nlev, ncol, 'cuda', seed=20250722, dtype=torch.float64)
e = 6.112 * torch.exp(17.67 * (td - 273.15) / (td - 29.65)); w = 0.6219569100577033 * e / (p - e); q = w / (1 + w)
z = 44330.8 * (1.0 - (p / 1013.25) ** 0.190263)
G = nlev * ncol * 8 / 1e9          # one grid in GB
def timed(f):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[2] * 1e3
prof = xa.cape_cin_columns(p, t, td, want_profile=True, moist='family')['profile']
sb = xa.cape_cin_columns(p, t, td, moist='family')
cases = {
 'dewpoint_from_specific_humidity': (lambda: xa.dewpoint_from_specific_humidity(p, t, q), 4 * G),
 'mixing_ratio': (lambda: xa.mixing_ratio(t, td, p), 4 * G),
 'most_unstable_parcel': (lambda: xa.most_unstable_parcel(p, t, td), 1.3 * G),
 'mixed_parcel': (lambda: xa.mixed_parcel(p, t, td), 0.8 * G),
 'lcl (surface)': (lambda: xa.lcl(p[0], t[0], td[0]), 0),
 'dry_lapse': (lambda: xa.dry_lapse(p, t[0]), 2 * G),
 'moist_lapse family': (lambda: xa.moist_lapse(p, t[0], moist='family'), 2 * G),
 'moist_lapse exact': (lambda: xa.moist_lapse(p, t[0], moist='exact'), 2 * G),
 'parcel_profile family': (lambda: xa.parcel_profile(p, p[0], t[0], td[0], moist='family'), 2 * G),
 'wet_bulb_temperature family': (lambda: xa.wet_bulb_temperature(p, t, td, moist='family'), 4 * G),
 'wet_bulb_temperature exact': (lambda: xa.wet_bulb_temperature(p, t, td, moist='exact'), 4 * G),
 'lfc_el': (lambda: xa.lfc_el(prof['pressure'], prof['temperature'], prof['environment_temperature'], sb['lcl_pressure'], sb['lcl_temperature']), 3 * G),
 'interp_level log': (lambda: xa.interp_level(p, t, 500.0, log=True), 2 * G),
 'crossing_level': (lambda: xa.crossing_level(z, t, 273.15), 2 * G),
 'find_intersections': (lambda: xa.find_intersections(p, t, td, log_x=True), 3 * G),
 'trapz': (lambda: xa.trapz(t, p), 2 * G),
 'get_layer': (lambda: xa.get_layer({'pressure': p, 'temperature': t, 'dewpoint': td}, depth=100), 3 * G),
 'mix_layer': (lambda: xa.mix_layer(p, t, td, depth=100), 6 * G),
 'from_most_unstable_parcel': (lambda: xa.from_most_unstable_parcel(p, t, td, depth=300), 6 * G),
}
out = {}
for k, (f, gb) in cases.items():
    try:
        ms = timed(f)
        out[k] = {'ms': round(ms, 3), 'min_GB': round(gb, 2), 'TBs': round(gb / ms, 2) if gb else None}
    except Exception as ex:
        out[k] = {'error': repr(ex)[:200]}
    print(k, out[k], flush=True)
