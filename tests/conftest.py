import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _native_builds():
    """Make sure the in-tree native pieces exist (no-ops when they are up to date): the HIP library is cross-compiled
    by hipcc, the C oracle by gcc.  Nothing here needs a GPU."""
    from xarray_parcel_amd import _lib
    _lib.build()
    from oracle import c_oracle
    c_oracle.build()
    yield


@pytest.fixture(autouse=True)
def _kat_moist_mode():
    """The reference runs its known-answer tests with parcel.moist_lapse replaced by MetPy's ODE
    (parcel_functions_demo.ipynb cell 33, unit_tests.py:114-140); the mirror's counterpart is set_moist_lapse('exact').
    Tests of the mirror's DEFAULT behaviour (lookup tables, as in the reference) switch it back with
    set_moist_lapse(None)."""
    from xarray_parcel_amd import parcel_functions as pf
    pf.set_moist_lapse('exact')
    yield
    pf.set_moist_lapse('exact')
