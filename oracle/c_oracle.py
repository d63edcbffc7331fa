"""
oracle/c_oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/c/xp_oracle.c.

Exposes (a) the grid entry point cape_cin_grid(), used by the differential tests, smoke()
and bench.py's cpu_baseline leg, and (b) the one-column `impl` API that
tests/kat_recipes.py drives (same function names as the reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libxp_oracle.so')
_lib = None

D = C.c_double
DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int32)


class Opts(C.Structure):
    _fields_ = [('vtc', C.c_int32), ('lcl_interp_log', C.c_int32), ('pos_cape_neg_cin', C.c_int32),
                ('post_zero_cin', C.c_int32), ('parcel_mode', C.c_int32), ('moist_mode', C.c_int32),
                ('depth', C.c_double)]


class Tables(C.Structure):
    _fields_ = [('n_p', C.c_int64), ('n_t', C.c_int64), ('n_adiabat', C.c_int64),
                ('p_max', D), ('p_step', D), ('t_min', D), ('t_step', D),
                ('index', C.POINTER(C.c_uint16)), ('adiabats', C.POINTER(C.c_float))]


def build(force=False):
    """Compile the C oracle (building the checker is not using it)."""
    src = os.path.join(_HERE, 'c', 'xp_oracle.c')
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, 'c', 'xp_oracle.c')
        if not os.path.exists(_LIB_PATH) or (os.path.exists(src) and os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
            build()                                        # never load a library older than its source
        _lib = C.CDLL(_LIB_PATH)
        _lib.xpo_mixed_layer.restype = D
    return _lib


def _a(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(DP)


PARCEL_MODES = {'surface': 0, 'most_unstable': 1, 'mixed_layer': 2, 'explicit': 3}
MOIST_MODES = {'rk4': 0, 'table': 1, 'family': 2}
_keep = []


def set_tables(tab):
    """tab: oracle.tables.Tables instance (index uint16 [n_p][n_t], adiabats float32 [n_ad][n_p] ascending p)."""
    t = Tables(tab.n_p, tab.n_t, tab.n_adiabat, tab.p_max, tab.p_step, tab.t_min, tab.t_step,
               tab.index.ctypes.data_as(C.POINTER(C.c_uint16)),
               tab.adiabats.ctypes.data_as(C.POINTER(C.c_float)))
    _keep.append((tab, t))
    lib().xpo_set_tables(C.byref(t))


def make_opts(virtual_temperature_correction=True, lcl_interp='log', pos_cape_neg_cin=True,
              post_zero_cin=False, parcel='surface', depth=None, moist='rk4'):
    if depth is None:
        depth = 300.0 if parcel == 'most_unstable' else 100.0
    return Opts(int(virtual_temperature_correction), int(lcl_interp == 'log'), int(pos_cape_neg_cin),
                int(post_zero_cin), PARCEL_MODES[parcel], MOIST_MODES[moist], float(depth))


SCALARS = ('cape', 'cin', 'lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature',
           'lfc_pressure', 'lfc_temperature', 'el_pressure', 'el_temperature')
INTS = ('lfc_index', 'el_index', 'status', 'parcel_index')
PROFILE = ('pressure', 'temperature', 'virtual_temperature', 'environment_temperature',
           'environment_virtual_temperature', 'environment_dewpoint')


def cape_cin_grid(p, t, td, parcel_values=None, want_profile=False, nthreads=0, **opts):
    """p, t, td: (nlev, ncol) arrays (any float dtype; computed in float64).  Returns dict of
    per-column outputs (+ 'profile' dict of (nlev+1, ncol) arrays if asked)."""
    p, t, td = _a(p), _a(t), _a(td)
    nlev, ncol = p.shape
    o = make_opts(**opts)
    sc = np.empty((9, ncol), dtype=np.float64)
    ints = np.empty((4, ncol), dtype=np.int32)
    prof = np.empty((6, nlev + 1, ncol), dtype=np.float64) if want_profile else None
    pv = None
    if parcel_values is not None:
        pv = _a(parcel_values)
        assert pv.shape == (3, ncol)
    lib().xpo_cape_cin(_p(p), _p(t), _p(td), C.c_int64(nlev), C.c_int64(ncol), C.c_int64(ncol), C.c_int64(1),
                       _p(pv) if pv is not None else None, C.byref(o), _p(sc), ints.ctypes.data_as(IP),
                       _p(prof) if prof is not None else None, C.c_int(nthreads))
    out = {k: sc[i] for i, k in enumerate(SCALARS)}
    out.update({k: ints[i] for i, k in enumerate(INTS)})
    if want_profile:
        out['profile'] = {k: prof[i] for i, k in enumerate(PROFILE)}
    return out


def family_table():
    """The C oracle's adiabat-family table as a (NPX, NDEG+1, MDEG+1, NPS) float64 array (see oracle/family.py)."""
    from . import family as fam
    lib().xpo_family_table.restype = DP
    ptr = lib().xpo_family_table()
    return np.ctypeslib.as_array(ptr, shape=(fam.NPX, fam.NDEG + 1, fam.MDEG + 1, fam.NPS)).copy()


def max_threads():
    return int(lib().xpo_max_threads())


# ---------------------------------------------------------------------------------------------
# one-column impl API for tests/kat_recipes.py (moist adiabat = RK4 spec unless set otherwise)
_MODE = {'moist': 'rk4'}


def set_moist_lapse(mode):
    assert mode in MOIST_MODES
    _MODE['moist'] = mode


def dry_lapse(pressure, parcel_temperature, parcel_pressure=None):
    # closed form; delegated to the array oracle's formula (pf.py:313) -- nothing to restate in C
    p = _a(pressure)
    if parcel_pressure is None:
        parcel_pressure = np.nanmax(p)
    return parcel_temperature * (p / parcel_pressure) ** (2.0 / 7.0)


def moist_lapse(pressure, parcel_temperature, parcel_pressure=None):
    p = _a(np.atleast_1d(pressure))
    if parcel_pressure is None:
        parcel_pressure = p[0]
    out = np.empty_like(p)
    lib().xpo_moist_lapse(C.c_int(len(p)), _p(p), D(parcel_temperature), D(parcel_pressure),
                          C.c_int(MOIST_MODES[_MODE['moist']]), _p(out))
    return out


def lcl(parcel_pressure, parcel_temperature, parcel_dewpoint):
    a, b, c = D(), D(), D()
    lib().xpo_lcl(D(parcel_pressure), D(parcel_temperature), D(parcel_dewpoint), C.byref(a), C.byref(b), C.byref(c))
    return {'lcl_pressure': a.value, 'lcl_temperature': b.value, 'lcl_virtual_temperature': c.value}


def parcel_profile(pressure, parcel_pressure, parcel_temperature, parcel_dewpoint):
    p = _a(pressure)
    n = len(p)
    tp, tvp, l3 = np.empty(n), np.empty(n), np.empty(3)
    lib().xpo_parcel_profile(C.c_int(n), _p(p), D(parcel_pressure), D(parcel_temperature), D(parcel_dewpoint),
                             C.c_int(MOIST_MODES[_MODE['moist']]), _p(tp), _p(tvp), _p(l3))
    return {'pressure': p, 'temperature': tp, 'virtual_temperature': tvp, 'lcl_pressure': l3[0],
            'lcl_temperature': l3[1], 'lcl_virtual_temperature': l3[2]}


def parcel_profile_with_lcl(pressure, temperature, dewpt, parcel_pressure, parcel_temperature, parcel_dewpoint,
                            lcl_interp='log'):
    p, t, td = _a(pressure), _a(temperature), _a(dewpt)
    n = len(p)
    out, l3 = np.empty((6, n + 1)), np.empty(3)
    lib().xpo_parcel_profile_with_lcl(C.c_int(n), _p(p), _p(t), _p(td), D(parcel_pressure), D(parcel_temperature),
                                      D(parcel_dewpoint), C.c_int(int(lcl_interp == 'log')),
                                      C.c_int(MOIST_MODES[_MODE['moist']]), _p(out), _p(l3))
    r = {k: out[i] for i, k in enumerate(PROFILE)}
    r.update({'lcl_pressure': l3[0], 'lcl_temperature': l3[1], 'lcl_virtual_temperature': l3[2]})
    return r


def lfc_el(pressure, parcel_temperature, temperature, lcl_pressure, lcl_temperature):
    p, par, env = _a(pressure), _a(parcel_temperature), _a(temperature)
    o4 = np.empty(4)
    i2 = np.empty(2, dtype=np.int32)
    st = lib().xpo_lfc_el(C.c_int(len(p)), _p(p), _p(par), _p(env), D(lcl_pressure), D(lcl_temperature), _p(o4),
                          i2.ctypes.data_as(IP))
    return {'lfc_pressure': o4[0], 'lfc_temperature': o4[1], 'el_pressure': o4[2], 'el_temperature': o4[3],
            'lfc_index': int(i2[0]), 'el_index': int(i2[1]), 'status_top_nan': bool(st & 1)}


def cape_cin_base(pressure, temperature, lfc_pressure, el_pressure, parcel_temperature,
                  pos_cape_neg_cin=True, post_zero_cin=False):
    p, env, par = _a(pressure), _a(temperature), _a(parcel_temperature)
    o2 = np.empty(2)
    lib().xpo_cape_cin_base(C.c_int(len(p)), _p(p), _p(env), D(lfc_pressure), D(el_pressure), _p(par),
                            C.c_int(int(pos_cape_neg_cin)), C.c_int(int(post_zero_cin)), _p(o2))
    return {'cape': o2[0], 'cin': o2[1]}


def _column(parcel, pressure, temperature, dewpt, depth=None, **kw):
    p, t, td = _a(pressure)[:, None], _a(temperature)[:, None], _a(dewpt)[:, None]
    r = cape_cin_grid(p, t, td, parcel=parcel, depth=depth, moist=_MODE['moist'], want_profile=True, **kw)
    cc = {'cape': float(r['cape'][0]), 'cin': float(r['cin'][0])}
    prof = {k: v[:, 0] for k, v in r['profile'].items()}
    for k in SCALARS[2:] + INTS:
        prof[k] = r[k][0]
    return cc, prof


def surface_based_cape_cin(pressure, temperature, dewpt, **kw):
    return _column('surface', pressure, temperature, dewpt, **kw)


def most_unstable_parcel(pressure, temperature, dewpt, depth=300):
    p, t, td = _a(pressure), _a(temperature), _a(dewpt)
    o3 = np.empty(3)
    idx = lib().xpo_most_unstable_parcel(C.c_int(len(p)), _p(p), _p(t), _p(td), D(depth), _p(o3))
    return {'pressure': o3[0], 'temperature': o3[1], 'dewpoint': o3[2], 'index': int(idx)}


def most_unstable_cape_cin(pressure, temperature, dewpt, depth=300, **kw):
    cc, prof = _column('most_unstable', pressure, temperature, dewpt, depth=depth, **kw)
    return cc, prof, most_unstable_parcel(pressure, temperature, dewpt, depth)


def mixed_layer(dat, depth=100):
    p = _a(dat['pressure'])
    return {k: float(lib().xpo_mixed_layer(C.c_int(len(p)), _p(p), _p(_a(v)), D(depth)))
            for k, v in dat.items() if k != 'pressure'}


def mixed_parcel(pressure, temperature, dewpt, depth=100):
    p, t, td = _a(pressure), _a(temperature), _a(dewpt)
    o3 = np.empty(3)
    lib().xpo_mixed_parcel(C.c_int(len(p)), _p(p), _p(t), _p(td), D(depth), _p(o3))
    return {'pressure': o3[0], 'temperature': o3[1], 'dewpoint': o3[2]}


def mixed_layer_cape_cin(pressure, temperature, dewpt, depth=100, **kw):
    cc, prof = _column('mixed_layer', pressure, temperature, dewpt, depth=depth, **kw)
    return cc, prof, mixed_parcel(pressure, temperature, dewpt, depth)
