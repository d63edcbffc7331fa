"""Several parcels of one grid in one pass (xp_cape_cin_multi, csrc/xp_multi.hpp): the fused kernel against separate
xp_cape_cin calls -- bit for bit, every output -- and against the oracle."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests.test_gpu_parity import MODES, _compare
from xarray_parcel_amd import synth

pytestmark = pytest.mark.gpu

SETS = [[('most_unstable', 300.0), ('mixed_layer', 100.0)],          # BASELINE config 5 (pf.py:1557, 1651)
        [('surface', None), ('most_unstable', 250.0)],
        [('mixed_layer', 100.0), ('mixed_layer', 50.0)],
        [('surface', None), ('surface', None)]]


@pytest.fixture(scope='module')
def xa():
    import torch
    assert torch.cuda.is_available(), 'these tests need the GPU'
    from xarray_parcel_amd import numpy_api
    return numpy_api


def _same(a, b, what):
    for k in b:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.dtype == y.dtype and x.shape == y.shape, (what, k)
        same = (x == y) | ((x != x) & (y != y))
        assert same.all(), (what, k, np.nonzero(~same)[0][:10], x[~same][:5], y[~same][:5])


@pytest.mark.parametrize('pset', range(len(SETS)))
@pytest.mark.parametrize('mode', range(len(MODES)))
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_fused_equals_separate_calls_bit_for_bit(xa, pset, mode, dtype):
    kw = MODES[mode]
    parcels = SETS[pset]
    for nlev, ncol, seed in ((48, 12000, 11 + mode), (9, 3000, 5), (100, 5000, 3)):
        p, t, td = synth.columns(nlev=nlev, ncol=ncol, seed=seed, nan_fraction=0.08, dtype=dtype)
        got = xa.cape_cin_multi(p, t, td, parcels, moist='family', fused=True, **kw)
        for (name, depth), g in zip(parcels, got):
            ref = xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, moist='family', **kw)
            _same(g, ref, (name, depth, nlev))


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_fused_vs_oracle(xa, dtype):
    p, t, td = synth.columns(nlev=64, ncol=20000, seed=23, nan_fraction=0.08, dtype=dtype)
    got = xa.cape_cin_multi(p, t, td, SETS[0], moist='family', fused=True)
    for (name, depth), g in zip(SETS[0], got):
        ref = co.cape_cin_grid(p, t, td, parcel=name, depth=depth, moist='family')
        _compare(g, ref, dtype, 1e-6)


def test_fused_truncated_and_ragged_shapes(xa):
    """1 ... 8 levels (LCL above the top, flush iteration), column counts around the wavefront / workgroup sizes, and a
    grid large enough for persistent wavefronts."""
    full = synth.columns(nlev=64, ncol=6000, seed=41, nan_fraction=0.08, dtype=np.float64)
    for nlev in (1, 2, 3, 5, 8):
        p, t, td = (np.ascontiguousarray(v[:nlev]) for v in full)
        got = xa.cape_cin_multi(p, t, td, SETS[0], moist='family', fused=True)
        for (name, depth), g in zip(SETS[0], got):
            _same(g, xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, moist='family'), (name, nlev))
    for ncol in (1, 63, 65, 511, 513, 1025):
        p, t, td = synth.columns(nlev=33, ncol=ncol, seed=ncol, nan_fraction=0.1, dtype=np.float32)
        got = xa.cape_cin_multi(p, t, td, SETS[0], moist='family', fused=True)
        for (name, depth), g in zip(SETS[0], got):
            _same(g, xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, moist='family'), (name, ncol))
    import torch
    p, t, td = synth.columns_torch(40, (1 << 19) + 777, 'cuda', seed=9, dtype=torch.float32)
    got = xa.cape_cin_multi(p, t, td, SETS[0], moist='family', fused=True, want=('cape', 'cin', 'lfc_index', 'el_index', 'parcel_index'))
    for (name, depth), g in zip(SETS[0], got):
        ref = xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, moist='family', want=('cape', 'cin', 'lfc_index', 'el_index', 'parcel_index'))
        for k in ref:
            assert torch.equal(g[k], ref[k]) or torch.equal(torch.isnan(g[k]), torch.isnan(ref[k])), (name, k)


def test_default_runs_the_parcels_one_after_the_other(xa):
    """Without fused=True, and for anything the fused kernel does not serve (moist='exact', three parcels), the same call
    answers with one pass per parcel."""
    p, t, td = synth.columns(nlev=30, ncol=3000, seed=2, nan_fraction=0.05, dtype=np.float64)
    for moist, fused in (('exact', True), ('family', False), ('family', True)):
        got = xa.cape_cin_multi(p, t, td, SETS[0] + [('surface', None)], moist=moist, fused=fused)
        for (name, depth), g in zip(SETS[0] + [('surface', None)], got):
            _same(g, xa.cape_cin_columns(p, t, td, parcel=name, depth=depth, moist=moist), (name,))
