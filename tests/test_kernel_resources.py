"""Build-time guard on the operating point of the headline kernel (no GPU needed: hipcc cross-compiles and reports the
resource usage): the surface-parcel CAPE/CIN instantiations must stay within 128 VGPRs -- four wavefronts per SIMD --
without spilling, and the workgroup's LDS (tables + per-thread scan slots) within a quarter of a CU's 160 KB.  The
kernel sits one or two registers under that limit, so an innocent edit can cost 13 %; this test says so."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which('hipcc')), reason='hipcc not available')
def test_surface_kernel_keeps_four_waves_per_simd(tmp_path):
    src = os.path.join(ROOT, 'xarray_parcel_amd', 'csrc', 'xp_cape_tu.hip')
    cmd = [HIPCC if os.path.exists(HIPCC) else 'hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-c',
           '-DXP_TU_T=double', '-DXP_TU_MODE=0', '-Rpass-analysis=kernel-resource-usage', '-o', str(tmp_path / 'tu.o'), src]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec, name = {}, None
    for ln in out.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', ln)
        if m:
            name = m.group(1)
            rec[name] = {}
        for key, pat in (('vgprs', r' VGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                         ('occupancy', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
            m = re.search(pat, ln)
            if m and name:
                rec[name][key] = int(m.group(1))
    # k_cape_cin<double, PM_SURFACE, PROFILE=false, MODE=0, HUM=false / true>
    for hum in ('0', '1'):
        k = [n for n in rec if re.search(r'k_cape_cinIdLi0ELb0ELi0ELb%sE' % hum, n)]
        assert len(k) == 1, list(rec)
        r = rec[k[0]]
        assert r['vgprs'] <= 128 and r['occupancy'] >= 4, r
        assert r['scratch'] <= 8, r                      # 8 B/lane is the call frame of the out-of-line slow paths
        assert r['lds'] <= 160 * 1024 // 4, r
