"""CPU-only checks of the drop-in boundary: the library loads and exports every symbol that
include/xparcel.h declares; the ctypes structures match the header's field order.  No compute."""
import ctypes as C
import os
import re

import pytest

from xarray_parcel_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, 'include', 'xparcel.h')).read()


def test_library_builds_and_loads():
    L.build()
    lib = L.load()
    assert lib.xp_version() == 100


def test_every_declared_symbol_is_exported():
    L.build()
    lib = L.load()
    declared = set(re.findall(r'^\s*(?:int|const char \*)\s*\*?\s*(xp_\w+)\s*\(', _header(), flags=re.M))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def _struct_fields(name):
    m = re.search(r'typedef struct \{([^{}]*)\} ' + name + ';', _header(), flags=re.S)
    body = re.sub(r'/\*.*?\*/', '', m.group(1), flags=re.S)
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(','):
            names.append(re.sub(r'\[.*\]', '', part.strip().split()[-1].lstrip('*')))
    return names


@pytest.mark.parametrize('cname,ctype', [('xp_view', L.View), ('xp_parcel', L.Parcel), ('xp_opts', L.Opts),
                                          ('xp_scalars_out', L.ScalarsOut), ('xp_profile_out', L.ProfileOut),
                                          ('xp_tables', L.Tables)])
def test_ctypes_structs_follow_header(cname, ctype):
    assert _struct_fields(cname) == [f[0] for f in ctype._fields_]


def test_no_gpu_means_loud_failure():
    """Without a device the product must refuse, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from xarray_parcel_amd import numpy_api as xa
    import numpy as np
    with pytest.raises(L.XParcelError):
        xa.lcl(1000.0, 300.0, 290.0)
    assert 'oracle' not in open(os.path.join(ROOT, 'xarray_parcel_amd', 'numpy_api.py')).read().replace('# oracle', '')
