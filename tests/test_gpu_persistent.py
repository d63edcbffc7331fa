"""Persistent wavefronts (family mode; the host picks them from 512 Ki / 4 Mi columns on): forced onto small grids through
XP_PERSIST_MIN_COLS = 0 -- read once per process, hence the subprocess -- and compared with the oracle like any other
launch, ragged column counts and grids smaller than one tile per wavefront included."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from oracle import c_oracle as co
from tests import test_gpu_parity as tp
from xarray_parcel_amd import numpy_api as xa, synth
tp.xa = xa
xa.set_family_table(co.family_table())
for ncol in (1, 63, 4097, 70001):
    for parcel in ('surface', 'most_unstable', 'mixed_layer'):
        for dtype in (np.float64, np.float32):
            p, t, td = synth.columns(nlev=33, ncol=ncol, seed=77 + ncol, nan_fraction=0.08, dtype=dtype)
            got = xa.cape_cin_columns(p, t, td, parcel=parcel, moist='family')
            ref = co.cape_cin_grid(p, t, td, parcel=parcel, moist='family')
            tp._compare(got, ref, dtype, 1e-6)
            lean = xa.cape_cin_columns(p, t, td, parcel=parcel, moist='family', want=('cape', 'cin'))
            assert np.array_equal(lean['cape'], got['cape'], equal_nan=True)
# specific-humidity input (the HUM instantiations) under persistence
from oracle import thermo as th
p, t, td = synth.columns(nlev=33, ncol=5000, seed=5, nan_fraction=0.08, dtype=np.float64)
e = th.saturation_vapor_pressure(td); w = th.EPSILON * e / (p - e); q = w / (1.0 + w)
with np.errstate(all='ignore'):
    td_ref = th.dewpoint_from_specific_humidity(p, t, q)
for parcel in ('surface', 'mixed_layer'):
    got = xa.cape_cin_columns(p, t, q, parcel=parcel, moist='family', humidity='specific')
    ref = co.cape_cin_grid(p, t, td_ref, parcel=parcel, moist='family')
    tp._compare(got, ref, np.float64, 1e-6)
print('PERSISTENT_OK')
''' % ROOT


@pytest.mark.gpu
def test_persistent_wavefronts_on_small_grids():
    env = dict(os.environ, XP_PERSIST_MIN_COLS='0')
    out = subprocess.run([sys.executable, '-c', SCRIPT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and 'PERSISTENT_OK' in out.stdout, out.stderr[-3000:]
