"""A small run of scripts/run_gpu_soak_profile.py: the six profile arrays of family mode (two-step Tv -> T inversion above the LCLs) against the C
oracle's, and the lifted-index-only kernels against the index of the written profile, over level counts x parcels x fp64 / fp32."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_profile_arrays_and_lifted_index_only_kernels_vs_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'run_gpu_soak_profile.py'), '6000', '2'], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    last = [l for l in out.stdout.splitlines() if l.startswith('PROFILE SOAK')][-1]
    assert '48 combinations x 6000 columns: 0 mismatching combinations' in last, last
