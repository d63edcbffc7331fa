"""
Array-level host API over libxparcel: the reference's function names
(modules/parcel_functions.py, "pf.py") on plain arrays.

Inputs are NumPy arrays (host memory, staged through the library) or torch CUDA
tensors (used in place, on torch's current stream), laid out (nlev, ...) with the
vertical first: (nlev,), (nlev, ncol) or (nlev, ny, nx).  Per-column results come
back with the horizontal shape of the input; profiles as (nlev+1, ...).  Returned
containers are plain dicts -- the xarray-facing mirrors in parcel_functions.py
wrap them into Datasets.

There is no CPU path here: every function ends in a kernel launch.
"""
import ctypes as C

import numpy as np

from . import _lib as L

L_EPS = 0.6219569100577033        # Mw / Md (metpy.constants, 1.4.1)

try:  # torch is plumbing (device memory, streams); the API also works without it on host arrays
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


import threading as _threading

_CTX = _threading.local()      # .device: the torch device of the call being assembled on this thread (None: host arrays)


def _ctx_device():
    return getattr(_CTX, 'device', None)


class _Arr:
    """Uniform handle on a NumPy array or torch CUDA tensor, flattened to (nlev, ncol)."""

    def __init__(self, x, dtype=None, like=None):
        if _is_torch(x):
            if not x.is_cuda:
                x = x.to(_ctx_device() or 'cuda')
            if dtype is not None:
                x = x.to(dtype=torch.float64 if dtype == np.float64 else torch.float32)
            elif x.dtype not in (torch.float32, torch.float64):
                x = x.to(torch.float64)
            self.t = x.contiguous()
            self.dev = True
            self.np_dtype = np.float64 if self.t.dtype == torch.float64 else np.float32
            self.shape = tuple(self.t.shape)
            self.ptr = self.t.data_ptr()
        else:
            a = np.asarray(x)
            if dtype is not None:
                a = a.astype(dtype, copy=False)
            elif a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            self.a = np.ascontiguousarray(a)
            self.dev = False
            self.np_dtype = self.a.dtype.type
            self.shape = self.a.shape
            self.ptr = self.a.ctypes.data

    @property
    def xp_dtype(self):
        return L.XP_F64 if self.np_dtype == np.float64 else L.XP_F32

    @property
    def mem(self):
        return L.XP_MEM_DEVICE if self.dev else L.XP_MEM_HOST


def _stream(dev):
    """torch's current stream on the device the call's tensors live on (not on torch's current device)."""
    if dev and torch is not None:
        return C.c_void_p(torch.cuda.current_stream(_ctx_device()).cuda_stream)
    return C.c_void_p(0)


def _device_of(h):
    if h.dev:
        return h.t.device.index
    return None


def _common(*xs):
    """Bring inputs to one dtype / one memory space; return handles + (nlev, ncol, hshape)."""
    devs = [x.device for x in xs if _is_torch(x) and x.is_cuda]
    any_dev = bool(devs)
    assert all(d == devs[0] for d in devs), 'all device tensors of one call must live on the same GPU'
    _CTX.device = devs[0] if any_dev else None
    f64 = any((_is_torch(x) and x.dtype == torch.float64) or
              (not _is_torch(x) and np.asarray(x).dtype != np.float32) for x in xs)
    dt = np.float64 if f64 else np.float32
    hs = []
    for x in xs:
        if any_dev and not _is_torch(x):
            x = torch.as_tensor(np.asarray(x, dtype=dt)).to(devs[0])
        hs.append(_Arr(x, dtype=dt))
    return hs, dt, any_dev


def _view(h, nlev, ncol):
    return L.View(h.ptr, h.xp_dtype, h.mem, nlev, ncol, ncol, 1)


def _alloc(shape, np_dtype, dev, like=None):
    if dev:
        td = {np.float64: torch.float64, np.float32: torch.float32, np.int32: torch.int32, np.uint8: torch.uint8}[np_dtype]
        t = torch.empty(shape, dtype=td, device=like.t.device if like is not None else 'cuda')
        return t, t.data_ptr()
    a = np.empty(shape, dtype=np_dtype)
    return a, a.ctypes.data


def _vert_shape(h):
    nlev = h.shape[0]
    hshape = tuple(h.shape[1:])
    ncol = int(np.prod(hshape)) if hshape else 1
    return nlev, ncol, hshape


def _per_col(x, ncol, dt, dev, like):
    """Per-column input (scalar or array of hshape) -> handle of ncol elements."""
    if _is_torch(x):
        h = _Arr((x.to(like.t.device) if dev else x).reshape(-1), dtype=dt)
    else:
        a = np.asarray(x, dtype=dt).reshape(-1)
        if a.size == 1 and ncol != 1:
            a = np.full(ncol, a[0], dtype=dt)
        h = _Arr(torch.as_tensor(a).to(like.t.device) if dev else a, dtype=dt)
    assert int(np.prod(h.shape)) == ncol, 'per-column argument does not match the grid'
    return h


def _opts(virtual_temperature_correction=True, lcl_interp='log', pos_cape_neg_cin=True, post_zero_cin=False,
          moist='exact', humidity='dewpoint'):
    if lcl_interp not in L.LCL_INTERP:
        raise AssertionError('interpolator must be linear or log')          # pf.py:878
    assert humidity in L.HUMIDITY, "humidity must be 'dewpoint' or 'specific'"
    return L.Opts(int(bool(virtual_temperature_correction)), L.LCL_INTERP[lcl_interp], int(bool(pos_cape_neg_cin)),
                  int(bool(post_zero_cin)), L.MOIST[moist], L.XP_F64, L.HUMIDITY[humidity], 0)


_DEFAULT = {'moist': 'exact'}


def set_moist_lapse(mode):
    """'exact' (RK4 integration of MetPy's ODE), 'family' (the same ODE from the adiabat-family table, faster) or
    'table' (the reference's lookup tables; needs adiabat_tables.load_moist_adiabat_lookups())."""
    assert mode in L.MOIST
    _DEFAULT['moist'] = mode


def cape_cin_columns(pressure, temperature, dewpoint, parcel='surface', depth=None, parcel_values=None,
                     want_profile=False, want=None, moist=None, lifted_index_at=None, **kwargs):
    """pf.py:1394-1475 with the three drivers.  Returns a dict of per-column arrays (and 'profile').
    want_profile: True for the six profile arrays of pf.py:806-931, or an iterable of their names for a subset (the ones
    not named are neither allocated nor written: lifted_index needs three of the six).
    lifted_index_at: a pressure [hPa] (the reference: 500): 'lifted_index' (pf.py:1722) of the lifted profile comes back
    with the scalars, computed in the same pass -- no profile array has to exist for it.
    humidity='specific' (keyword): `dewpoint` holds specific humidity [kg/kg] and is converted on load
    (parcel_test.py:262-266 fused into the pass)."""
    (p, t, td), dt, dev = _common(pressure, temperature, dewpoint)
    assert p.shape == t.shape == td.shape, 'pressure, temperature, dewpoint must share a shape'
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    o = _opts(moist=moist or _DEFAULT['moist'], **kwargs)
    if depth is None:
        depth = 300.0 if parcel == 'most_unstable' else 100.0                # pf.py:1558, 1652
    pc = L.Parcel(L.PARCEL[parcel], 0, float(depth), None, None, None)
    keep = []
    if parcel == 'explicit':
        hs = [_per_col(x, ncol, dt, dev, p) for x in parcel_values]
        keep += hs
        pc.pressure, pc.temperature, pc.dewpoint = hs[0].ptr, hs[1].ptr, hs[2].ptr
    names = L.SCALAR_F + L.SCALAR_I + L.SCALAR_P if want is None else tuple(want)
    so = L.ScalarsOut()
    so.dtype = p.xp_dtype
    so.mem = p.mem
    out = {}
    for k in names:
        arr, ptr = _alloc((ncol,), np.int32 if k in L.SCALAR_I else dt, dev, p)
        setattr(so, k, ptr)
        out[k] = arr
    po = None
    if want_profile or lifted_index_at is not None:
        po = L.ProfileOut()
        po.dtype, po.mem, po.nlev_out, po.lev_stride, po.col_stride = p.xp_dtype, p.mem, nlev + 1, ncol, 1
        prof = {}
        if lifted_index_at is not None:
            arr, ptr = _alloc((ncol,), dt, dev, p)
            po.lifted_index, po.lifted_index_pressure = ptr, float(lifted_index_at)
            out['lifted_index'] = arr
        pvars = L.PROFILE_VARS if want_profile is True else tuple(want_profile or ())
        assert all(k in L.PROFILE_VARS for k in pvars), f'profile variables are {L.PROFILE_VARS}'
        for k in pvars:
            arr, ptr = _alloc((nlev + 1, ncol), dt, dev, p)
            setattr(po, k, ptr)
            prof[k] = arr
    L.check(lib.xp_cape_cin(C.byref(_view(p, nlev, ncol)), C.byref(_view(t, nlev, ncol)),
                            C.byref(_view(td, nlev, ncol)), C.byref(pc), C.byref(o), C.byref(so),
                            C.byref(po) if po is not None else None, _stream(dev)))
    res = {k: v.reshape(hshape) for k, v in out.items()}
    if want_profile:
        res['profile'] = {k: v.reshape((nlev + 1,) + hshape) for k, v in prof.items()}
    return res


def cape_cin_multi(pressure, temperature, dewpoint, parcels, want=None, moist=None, lifted_index_at=None, fused=False, **kwargs):
    """Several parcels of one grid in ONE call (xp_cape_cin_multi): `parcels` is a sequence of names or (name, depth)
    pairs out of 'surface', 'most_unstable', 'mixed_layer' -- e.g. [('most_unstable', 300), ('mixed_layer', 100)] for
    BASELINE config 5, [('most_unstable', 250), ('mixed_layer', 100), ('mixed_layer', 50)] for conv_properties
    (pf.py:1984-2006).  Returns one dict per parcel, bit-identical to what cape_cin_columns() returns for it.
    fused=True (XP_OPT_FUSE_PARCELS; moist='family', two parcels): one pass over the grid for both -- same numbers,
    measured slower than a pass per parcel on MI355X, hence opt-in."""
    (p, t, td), dt, dev = _common(pressure, temperature, dewpoint)
    assert p.shape == t.shape == td.shape, 'pressure, temperature, dewpoint must share a shape'
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    o = _opts(moist=moist or _DEFAULT['moist'], **kwargs)
    o.flags = L.OPT_FUSE_PARCELS if fused else 0
    specs = []
    for pc in parcels:
        name, depth = (pc, None) if isinstance(pc, str) else (pc[0], pc[1])
        assert name in ('surface', 'most_unstable', 'mixed_layer'), 'parcels: surface, most_unstable or mixed_layer'
        if depth is None:
            depth = 300.0 if name == 'most_unstable' else 100.0                # pf.py:1558, 1652
        specs.append((name, float(depth)))
    n = len(specs)
    pcs = (L.Parcel * n)(*[L.Parcel(L.PARCEL[nm], 0, dp, None, None, None) for nm, dp in specs])
    sos = (L.ScalarsOut * n)()
    pos = (L.ProfileOut * n)() if lifted_index_at is not None else None
    names = L.SCALAR_F + L.SCALAR_I + L.SCALAR_P if want is None else tuple(want)
    outs = []
    for i in range(n):
        sos[i].dtype, sos[i].mem = p.xp_dtype, p.mem
        out = {}
        for k in names:
            arr, ptr = _alloc((ncol,), np.int32 if k in L.SCALAR_I else dt, dev, p)
            setattr(sos[i], k, ptr)
            out[k] = arr
        if pos is not None:
            pos[i].dtype, pos[i].mem, pos[i].nlev_out, pos[i].lev_stride, pos[i].col_stride = p.xp_dtype, p.mem, nlev + 1, ncol, 1
            arr, ptr = _alloc((ncol,), dt, dev, p)
            pos[i].lifted_index, pos[i].lifted_index_pressure = ptr, float(lifted_index_at)
            out['lifted_index'] = arr
        outs.append(out)
    L.check(lib.xp_cape_cin_multi(C.byref(_view(p, nlev, ncol)), C.byref(_view(t, nlev, ncol)), C.byref(_view(td, nlev, ncol)),
                                  C.c_int32(n), pcs, C.byref(o), sos, pos, _stream(dev)))
    return [{k: v.reshape(hshape) for k, v in out.items()} for out in outs]


# ---- reference-named functions (one column or a grid) -----------------------------------------
def _split(res):
    cc = {'cape': res['cape'], 'cin': res['cin']}
    prof = dict(res.get('profile', {}))
    for k in ('lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature', 'lfc_pressure', 'lfc_temperature',
              'el_pressure', 'el_temperature', 'lfc_index', 'el_index', 'status'):
        prof[k] = res[k]
    return cc, prof


def cape_cin(pressure, temperature, dewpoint, parcel_temperature, parcel_pressure, parcel_dewpoint, **kwargs):
    """pf.py:1394."""
    return _split(cape_cin_columns(pressure, temperature, dewpoint, parcel='explicit',
                                   parcel_values=(parcel_pressure, parcel_temperature, parcel_dewpoint),
                                   want_profile=True, **kwargs))


def surface_based_cape_cin(pressure, temperature, dewpoint, **kwargs):
    """pf.py:1477."""
    return _split(cape_cin_columns(pressure, temperature, dewpoint, parcel='surface', want_profile=True, **kwargs))


def most_unstable_cape_cin(pressure, temperature, dewpoint, depth=300, **kwargs):
    """pf.py:1557."""
    res = cape_cin_columns(pressure, temperature, dewpoint, parcel='most_unstable', depth=depth, want_profile=True,
                           **kwargs)
    cc, prof = _split(res)
    return cc, prof, {'pressure': res['parcel_pressure'], 'temperature': res['parcel_temperature'],
                      'dewpoint': res['parcel_dewpoint'], 'index': res['parcel_index']}


def mixed_layer_cape_cin(pressure, temperature, dewpoint, depth=100, **kwargs):
    """pf.py:1651."""
    res = cape_cin_columns(pressure, temperature, dewpoint, parcel='mixed_layer', depth=depth, want_profile=True,
                           **kwargs)
    cc, prof = _split(res)
    return cc, prof, {'pressure': res['parcel_pressure'], 'temperature': res['parcel_temperature'],
                      'dewpoint': res['parcel_dewpoint']}


def parcel_profile_with_lcl(pressure, temperature, dewpoint, parcel_pressure, parcel_temperature, parcel_dewpoint,
                            lcl_interp='log', moist=None):
    """pf.py:806."""
    res = cape_cin_columns(pressure, temperature, dewpoint, parcel='explicit',
                           parcel_values=(parcel_pressure, parcel_temperature, parcel_dewpoint), want_profile=True,
                           lcl_interp=lcl_interp, moist=moist,
                           want=('lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature'))
    out = dict(res['profile'])
    for k in ('lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature'):
        out[k] = res[k]
    return out


def _select(pressure, temperature, dewpoint, mode, depth):
    (p, t, td), dt, dev = _common(pressure, temperature, dewpoint)
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    pc = L.Parcel(L.PARCEL[mode], 0, float(depth), None, None, None)
    so = L.ScalarsOut()
    so.dtype, so.mem = p.xp_dtype, p.mem
    out = {}
    for k in L.SCALAR_P + ('parcel_index',):
        arr, ptr = _alloc((ncol,), np.int32 if k == 'parcel_index' else dt, dev, p)
        setattr(so, k, ptr)
        out[k] = arr
    L.check(lib.xp_select_parcel(C.byref(_view(p, nlev, ncol)), C.byref(_view(t, nlev, ncol)),
                                 C.byref(_view(td, nlev, ncol)), C.byref(pc), C.byref(so), _stream(dev)))
    return {'pressure': out['parcel_pressure'].reshape(hshape), 'temperature': out['parcel_temperature'].reshape(hshape),
            'dewpoint': out['parcel_dewpoint'].reshape(hshape), 'index': out['parcel_index'].reshape(hshape)}


def most_unstable_parcel(pressure, temperature, dewpoint, depth=300):
    """pf.py:102."""
    return _select(pressure, temperature, dewpoint, 'most_unstable', depth)


def mixed_parcel(pressure, temperature, dewpoint, depth=100):
    """pf.py:229."""
    r = _select(pressure, temperature, dewpoint, 'mixed_layer', depth)
    r.pop('index')
    return r


def mixed_layer(dat, depth=100):
    """pf.py:137: dat = dict with 'pressure' and variables to mix."""
    out = {}
    for k, v in dat.items():
        if k == 'pressure':
            continue
        (p, x), dt, dev = _common(dat['pressure'], v)
        nlev, ncol, hshape = _vert_shape(p)
        lib = L.init(_device_of(p))
        arr, ptr = _alloc((ncol,), dt, dev, p)
        L.check(lib.xp_mixed_layer(C.byref(_view(p, nlev, ncol)), C.byref(_view(x, nlev, ncol)), C.c_double(depth),
                                   C.c_void_p(ptr), _stream(dev)))
        out[k] = arr.reshape(hshape)
    return out


def lcl(parcel_pressure, parcel_temperature, parcel_dewpoint):
    """pf.py:609."""
    (p, t, td), dt, dev = _common(*(x if _is_torch(x) else np.atleast_1d(np.asarray(x, dtype=np.float64))
                                    for x in (parcel_pressure, parcel_temperature, parcel_dewpoint)))
    n = int(np.prod(p.shape))
    shape = tuple(p.shape) if np.ndim(parcel_pressure) else ()
    lib = L.init(_device_of(p))
    outs = [_alloc((n,), dt, dev, p) for _ in range(3)]
    st, stp = _alloc((n,), np.int32, dev, p)
    L.check(lib.xp_lcl(C.c_int64(n), p.xp_dtype, p.mem, C.c_void_p(p.ptr), C.c_void_p(t.ptr), C.c_void_p(td.ptr),
                       C.c_void_p(outs[0][1]), C.c_void_p(outs[1][1]), C.c_void_p(outs[2][1]), C.c_void_p(stp),
                       _stream(dev)))
    return {'lcl_pressure': outs[0][0].reshape(shape), 'lcl_temperature': outs[1][0].reshape(shape),
            'lcl_virtual_temperature': outs[2][0].reshape(shape)}


def _lapse(fn_name, pressure, parcel_temperature, parcel_pressure, moist_mode=None):
    (p,), dt, dev = _common(pressure)
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    pt = _per_col(parcel_temperature, ncol, dt, dev, p)
    pp = _per_col(parcel_pressure, ncol, dt, dev, p) if parcel_pressure is not None else None
    out, optr = _alloc((nlev, ncol), dt, dev, p)
    args = [C.byref(_view(p, nlev, ncol)), C.c_void_p(pt.ptr), C.c_void_p(pp.ptr) if pp is not None else None]
    if moist_mode is not None:
        args.append(C.c_int32(moist_mode))
    args += [C.c_void_p(optr), _stream(dev)]
    L.check(getattr(lib, fn_name)(*args))
    return out.reshape((nlev,) + hshape)


def dry_lapse(pressure, parcel_temperature, parcel_pressure=None):
    """pf.py:291."""
    return _lapse('xp_dry_lapse', pressure, parcel_temperature, parcel_pressure)


def moist_lapse(pressure, parcel_temperature, parcel_pressure=None, moist=None):
    """pf.py:525."""
    return _lapse('xp_moist_lapse', pressure, parcel_temperature, parcel_pressure,
                  moist_mode=L.MOIST[moist or _DEFAULT['moist']])


def parcel_profile(pressure, parcel_pressure, parcel_temperature, parcel_dewpoint, moist=None):
    """pf.py:712."""
    (p,), dt, dev = _common(pressure)
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    pp, pt, ptd = (_per_col(x, ncol, dt, dev, p) for x in (parcel_pressure, parcel_temperature, parcel_dewpoint))
    t_out, tptr = _alloc((nlev, ncol), dt, dev, p)
    tv_out, tvptr = _alloc((nlev, ncol), dt, dev, p)
    ls = [_alloc((ncol,), dt, dev, p) for _ in range(3)]
    L.check(lib.xp_parcel_profile(C.byref(_view(p, nlev, ncol)), C.c_void_p(pp.ptr), C.c_void_p(pt.ptr),
                                  C.c_void_p(ptd.ptr), C.c_int32(L.MOIST[moist or _DEFAULT['moist']]),
                                  C.c_void_p(tptr), C.c_void_p(tvptr), C.c_void_p(ls[0][1]), C.c_void_p(ls[1][1]),
                                  C.c_void_p(ls[2][1]), _stream(dev)))
    pres = p.t if dev else p.a
    return {'pressure': pres, 'temperature': t_out.reshape((nlev,) + hshape),
            'virtual_temperature': tv_out.reshape((nlev,) + hshape), 'lcl_pressure': ls[0][0].reshape(hshape),
            'lcl_temperature': ls[1][0].reshape(hshape), 'lcl_virtual_temperature': ls[2][0].reshape(hshape)}


def lfc_el(pressure, parcel_temperature, temperature, lcl_pressure, lcl_temperature):
    """pf.py:1066."""
    (p, par, env), dt, dev = _common(pressure, parcel_temperature, temperature)
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    lp, lt = (_per_col(x, ncol, dt, dev, p) for x in (lcl_pressure, lcl_temperature))
    so = L.ScalarsOut()
    so.dtype, so.mem = p.xp_dtype, p.mem
    out = {}
    for k in ('lfc_pressure', 'lfc_temperature', 'el_pressure', 'el_temperature', 'lfc_index', 'el_index', 'status'):
        arr, ptr = _alloc((ncol,), np.int32 if k in L.SCALAR_I else dt, dev, p)
        setattr(so, k, ptr)
        out[k] = arr.reshape(hshape)
    L.check(lib.xp_lfc_el(C.byref(_view(p, nlev, ncol)), C.byref(_view(par, nlev, ncol)),
                          C.byref(_view(env, nlev, ncol)), C.c_void_p(lp.ptr), C.c_void_p(lt.ptr), C.byref(so),
                          _stream(dev)))
    return out


def cape_cin_base(pressure, temperature, lfc_pressure, el_pressure, parcel_temperature, pos_cape_neg_cin=True,
                  post_zero_cin=False, **_ignored):
    """pf.py:1291."""
    (p, env, par), dt, dev = _common(pressure, temperature, parcel_temperature)
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    lf, el = (_per_col(x, ncol, dt, dev, p) for x in (lfc_pressure, el_pressure))
    o = _opts(pos_cape_neg_cin=pos_cape_neg_cin, post_zero_cin=post_zero_cin)
    cape, cptr = _alloc((ncol,), dt, dev, p)
    cin, nptr = _alloc((ncol,), dt, dev, p)
    L.check(lib.xp_cape_cin_base(C.byref(_view(p, nlev, ncol)), C.byref(_view(env, nlev, ncol)),
                                 C.byref(_view(par, nlev, ncol)), C.c_void_p(lf.ptr), C.c_void_p(el.ptr), C.byref(o),
                                 C.c_void_p(cptr), C.c_void_p(nptr), _stream(dev)))
    return {'cape': cape.reshape(hshape), 'cin': cin.reshape(hshape)}


# ---- SURVEY 8(f) items that reuse the hot-path device code -------------------------------------------------
def wet_bulb_temperature(pressure, temperature, dewpoint, moist=None):
    """pf.py:389: Normand's rule, every element independently."""
    (p, t, td), dt, dev = _common(pressure, temperature, dewpoint)
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    out, optr = _alloc((nlev, ncol), dt, dev, p)
    L.check(lib.xp_wet_bulb_temperature(C.byref(_view(p, nlev, ncol)), C.byref(_view(t, nlev, ncol)),
                                        C.byref(_view(td, nlev, ncol)), C.c_int32(L.MOIST[moist or _DEFAULT['moist']]),
                                        C.c_void_p(optr), _stream(dev)))
    return out.reshape((nlev,) + hshape)


def interp_level(coords, variable, at, log=False):
    """pf.py:1758 linear_interp (log=False) / pf.py:1813 log_interp (log=True) of one variable."""
    (cds, x), dt, dev = _common(coords, variable)
    nlev, ncol, hshape = _vert_shape(cds)
    lib = L.init(_device_of(cds))
    scalar = np.ndim(at) == 0 and not _is_torch(at)
    ah = _Arr(torch.as_tensor(np.asarray([at], dtype=dt)).to(cds.t.device) if dev else np.asarray([at], dtype=dt),
              dtype=dt) if scalar else _per_col(at, ncol, dt, dev, cds)
    out, optr = _alloc((ncol,), dt, dev, cds)
    L.check(lib.xp_interp_level(C.byref(_view(cds, nlev, ncol)), C.byref(_view(x, nlev, ncol)), C.c_void_p(ah.ptr),
                                C.c_int32(int(scalar)), C.c_int32(int(log)), C.c_void_p(optr), _stream(dev)))
    return out.reshape(hshape)


def interp_levels(coords, variables, ats, log=False):
    """interp_level() for up to four variables at up to four scalar coordinates in one pass over the column
    (xp_interp_levels): returns [[variable v at ats[j] for j] for v]."""
    arrs, dt, dev = _common(coords, *variables)
    cds, xs = arrs[0], arrs[1:]
    assert 1 <= len(xs) <= 4 and 1 <= len(ats) <= 4, 'one to four variables, one to four coordinates'
    assert all(x.shape == cds.shape for x in xs), 'coords and variables must share a shape'
    nlev, ncol, hshape = _vert_shape(cds)
    lib = L.init(_device_of(cds))
    views = [_view(x, nlev, ncol) for x in xs]
    vptrs = (C.POINTER(L.View) * len(xs))(*[C.pointer(v) for v in views])
    outs = [[_alloc((ncol,), dt, dev, cds) for _ in ats] for _ in xs]
    optrs = (C.c_void_p * (len(xs) * len(ats)))(*[o[1] for row in outs for o in row])
    at = (C.c_double * len(ats))(*[float(a) for a in ats])
    L.check(lib.xp_interp_levels(C.byref(_view(cds, nlev, ncol)), C.c_int32(len(xs)), vptrs, C.c_int32(len(ats)), at,
                                 C.c_int32(int(log)), optrs, _stream(dev)))
    return [[o[0].reshape(hshape) for o in row] for row in outs]


def dewpoint_from_specific_humidity(pressure, temperature, specific_humidity):
    """metpy.calc.dewpoint_from_specific_humidity, MetPy 1.4.1 chain (parcel_test.py:262-266, pf.py:1889), K."""
    (p, t, q), dt, dev = _common(pressure, temperature, specific_humidity)
    assert p.shape == t.shape == q.shape, 'pressure, temperature, specific_humidity must share a shape'
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    out, optr = _alloc((nlev, ncol), dt, dev, p)
    L.check(lib.xp_dewpoint_from_specific_humidity(C.byref(_view(p, nlev, ncol)), C.byref(_view(t, nlev, ncol)),
                                                   C.byref(_view(q, nlev, ncol)), C.c_void_p(optr), _stream(dev)))
    return out.reshape((nlev,) + hshape)


def mixing_ratio(temperature, dewpoint, pressure):
    """pf.py:684: RH(T, Td) x saturation mixing ratio at (p, T) [kg/kg]."""
    (t, td, p), dt, dev = _common(temperature, dewpoint, pressure)
    assert t.shape == td.shape == p.shape, 'temperature, dewpoint, pressure must share a shape'
    shape = t.shape
    n = int(np.prod(shape)) if shape else 1
    lib = L.init(_device_of(t))
    out, optr = _alloc((n,), dt, dev, t)
    L.check(lib.xp_mixing_ratio(C.byref(_view(t, 1, n)), C.byref(_view(td, 1, n)), C.byref(_view(p, 1, n)),
                                C.c_void_p(optr), _stream(dev)))
    return out.reshape(shape)


def virtual_temperature(temperature, mixing_ratio, epsilon=0.608):
    """pf.py:782 (Doswell & Rasmussen 1994): one multiply-add, plain array arithmetic."""
    return temperature * (1 + epsilon * mixing_ratio)


def crossing_level(x, a, value):
    """Smallest x over all intersections of the profile a(x) with the constant `value` (find_intersections
    pf.py:992 + the min of pf.py:2153): freezing_level_height is crossing_level(height, temperature, 273.15)."""
    (xh, ah), dt, dev = _common(x, a)
    assert xh.shape == ah.shape
    nlev, ncol, hshape = _vert_shape(xh)
    lib = L.init(_device_of(xh))
    out, optr = _alloc((ncol,), dt, dev, xh)
    L.check(lib.xp_crossing_level(C.byref(_view(xh, nlev, ncol)), C.byref(_view(ah, nlev, ncol)), C.c_double(float(value)),
                                  C.c_void_p(optr), _stream(dev)))
    return out.reshape(hshape)



# -- the reference's array primitives (pf.py:63-100, 164-227, 858-1064, 1200-1289, 1517-1555, 1604-1649, 1699-1720) --------
# The CAPE / CIN kernels never build these arrays; the functions exist because the reference offers them to its callers.
# Arrays are (nlev, ...) with the vertical first, datasets are dicts name -> array.
def insert_level(d, level, coords='pressure', fill_value=-999):
    """pf.py:933: insert `level` (dict name -> one value per column) into the dataset `d` sorted by decreasing `coords`;
    the keys of `level` define the output (pf.py:983)."""
    out = {}
    for k in level.keys():
        (cds, v), dt, dev = _common(d[coords], d[k])
        assert cds.shape == v.shape, 'variables of a dataset must share a shape'
        nlev, ncol, hshape = _vert_shape(cds)
        lib = L.init(_device_of(cds))
        lc, lv = _per_col(level[coords], ncol, dt, dev, cds), _per_col(level[k], ncol, dt, dev, cds)
        arr, ptr = _alloc((nlev + 1, ncol), dt, dev, cds)
        L.check(lib.xp_insert_level(C.byref(_view(cds, nlev, ncol)), C.byref(_view(v, nlev, ncol)), C.c_void_p(lc.ptr),
                                    C.c_void_p(lv.ptr), C.c_double(float(fill_value)), C.c_void_p(ptr), _stream(dev)))
        out[k] = arr.reshape((nlev + 1,) + hshape)
    return out


INTERSECTION_KEYS = ('all_intersect_x', 'all_intersect_y', 'increasing_x', 'increasing_y', 'decreasing_x', 'decreasing_y')


def find_intersections(x, a, b=None, log_x=False):
    """pf.py:992: dict of six (nlev - 1, ...) arrays; entry i belongs to the interval between levels i and i + 1."""
    ins = (x, a) if b is None else (x, a, b)
    hs, dt, dev = _common(*ins)
    xh = hs[0]
    assert all(h.shape == xh.shape for h in hs), 'x, a, b must share a shape'
    nlev, ncol, hshape = _vert_shape(xh)
    lib = L.init(_device_of(xh))
    outs = [_alloc((nlev - 1, ncol), dt, dev, xh) for _ in INTERSECTION_KEYS]
    optrs = (C.c_void_p * 6)(*[o[1] for o in outs])
    bview = C.byref(_view(hs[2], nlev, ncol)) if b is not None else None
    L.check(lib.xp_find_intersections(C.byref(_view(xh, nlev, ncol)), C.byref(_view(hs[1], nlev, ncol)), bview,
                                      C.c_int32(int(bool(log_x))), optrs, _stream(dev)))
    return {k: o[0].reshape((nlev - 1,) + hshape) for k, o in zip(INTERSECTION_KEYS, outs)}


def _mask_handle(mask, shape, dev, like):
    """bool / integer mask -> dense uint8 of `shape` in the memory space of the call."""
    if _is_torch(mask):
        if dev:
            m = (mask != 0).to(torch.uint8).reshape(shape).contiguous().to(like.t.device)
            return m, m.data_ptr()
        mask = mask.cpu().numpy()
    m = np.ascontiguousarray((np.asarray(mask) != 0).reshape(shape), dtype=np.uint8)
    if dev:
        m = torch.as_tensor(m).to(like.t.device)
        return m, m.data_ptr()
    return m, m.ctypes.data


def trapz(dat, x, mask=None, only_positive=False, only_negative=False):
    """pf.py:164 for one variable (array) or several (dict name -> array): sum of |dx| * mean over the intervals."""
    assert not (only_positive and only_negative), 'Only negative OR positive regions can be included in trapz.'   # pf.py:200
    if isinstance(dat, dict):
        return {k: trapz(v, x, mask=mask, only_positive=only_positive, only_negative=only_negative) for k, v in dat.items()}
    (d, xh), dt, dev = _common(dat, x)
    assert d.shape == xh.shape, 'dat and x must share a shape'
    nlev, ncol, hshape = _vert_shape(d)
    lib = L.init(_device_of(d))
    keep, mptr = (None, None) if mask is None else _mask_handle(mask, (max(nlev - 1, 0), ncol), dev, d)
    out, optr = _alloc((ncol,), dt, dev, d)
    L.check(lib.xp_trapz(C.byref(_view(d, nlev, ncol)), C.byref(_view(xh, nlev, ncol)), C.c_void_p(mptr),
                         C.c_int32(int(bool(only_positive))), C.c_int32(int(bool(only_negative))), C.c_void_p(optr), _stream(dev)))
    del keep
    return out.reshape(hshape)


AREA_KEYS = ('area', 'dx', 'x', 'x_from', 'x_to')


def trap_around_zeros(x, y, log_x=True, start=0):
    """pf.py:1200 (start = 0, the only value the reference uses): (areas, mask).  areas: dict of five (2 nlev - 1, ...)
    arrays, the areas before the zeros of y (rows 0 .. nlev-1) followed by the areas after them; mask: (nlev, ...) bool,
    True where no area was taken out of an interval."""
    assert start == 0, 'only start=0 is implemented (the reference never passes anything else)'
    (xh, yh), dt, dev = _common(x, y)
    assert xh.shape == yh.shape, 'x and y must share a shape'
    nlev, ncol, hshape = _vert_shape(xh)
    lib = L.init(_device_of(xh))
    outs = [_alloc((2 * nlev - 1, ncol), dt, dev, xh) for _ in AREA_KEYS]
    optrs = (C.c_void_p * 5)(*[o[1] for o in outs])
    mask, mptr = _alloc((nlev, ncol), np.uint8, dev, xh)
    L.check(lib.xp_trap_around_zeros(C.byref(_view(xh, nlev, ncol)), C.byref(_view(yh, nlev, ncol)), C.c_int32(int(bool(log_x))),
                                     optrs, C.c_void_p(mptr), _stream(dev)))
    areas = {k: o[0].reshape((2 * nlev - 1,) + hshape) for k, o in zip(AREA_KEYS, outs)}
    return areas, (mask != 0).reshape((nlev,) + hshape)


def bound_pressure(pressure, bound):
    """pf.py:208: the pressure of each column closest to `bound` (scalar or one per column)."""
    (p,), dt, dev = _common(pressure)
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    b = _per_col(bound, ncol, dt, dev, p)
    out, optr = _alloc((ncol,), dt, dev, p)
    L.check(lib.xp_bound_pressure(C.byref(_view(p, nlev, ncol)), C.c_void_p(b.ptr), C.c_void_p(optr), _stream(dev)))
    return out.reshape(hshape)


def get_layer(dat, depth=100, interpolate=True):
    """pf.py:63: dat = dict with 'pressure' and variables; the lowest `depth` hPa, NaN outside; with interpolate the layer top
    is inserted as a level (nlev + 1 rows)."""
    out = {}
    for k, v in dat.items():
        (p, x), dt, dev = _common(dat['pressure'], v)
        assert p.shape == x.shape, 'variables of a dataset must share a shape'
        nlev, ncol, hshape = _vert_shape(p)
        lib = L.init(_device_of(p))
        rows = nlev + (1 if interpolate else 0)
        arr, ptr = _alloc((rows, ncol), dt, dev, p)
        L.check(lib.xp_get_layer(C.byref(_view(p, nlev, ncol)), C.byref(_view(x, nlev, ncol)), C.c_double(float(depth)),
                                 C.c_int32(int(bool(interpolate))), C.c_int32(int(k == 'pressure')), C.c_void_p(ptr), _stream(dev)))
        out[k] = arr.reshape((rows,) + hshape)
    return out


def shift_out_nans(x, name):
    """pf.py:1699: x = dict name -> array; every column of every variable is moved down by the number of leading NaNs of
    x[name] in that column."""
    out = {}
    for k, v in x.items():
        (nh, vh), dt, dev = _common(x[name], v)
        assert nh.shape == vh.shape, 'variables of a dataset must share a shape'
        nlev, ncol, hshape = _vert_shape(nh)
        lib = L.init(_device_of(nh))
        arr, ptr = _alloc((nlev, ncol), dt, dev, nh)
        L.check(lib.xp_shift_out_nans(C.byref(_view(nh, nlev, ncol)), C.byref(_view(vh, nlev, ncol)), C.c_void_p(ptr), _stream(dev)))
        out[k] = arr.reshape((nlev,) + hshape)
    return out


def _rebase(pressure, temperature, dewpoint, mode, depth):
    (p, t, td), dt, dev = _common(pressure, temperature, dewpoint)
    assert p.shape == t.shape == td.shape, 'pressure, temperature, dewpoint must share a shape'
    nlev, ncol, hshape = _vert_shape(p)
    lib = L.init(_device_of(p))
    pc = L.Parcel(L.PARCEL[mode], 0, float(depth), None, None, None)
    so = L.ScalarsOut()
    so.dtype, so.mem = p.xp_dtype, p.mem
    par = {}
    for k in L.SCALAR_P + ('parcel_index',):
        arr, ptr = _alloc((ncol,), np.int32 if k == 'parcel_index' else dt, dev, p)
        setattr(so, k, ptr)
        par[k] = arr
    rows = nlev + (1 if mode == 'mixed_layer' else 0)
    outs = [_alloc((rows, ncol), dt, dev, p) for _ in range(3)]
    kept = np.zeros(nlev, dtype=np.int32)
    nout = C.c_int64(0)
    L.check(lib.xp_rebase_profile(C.byref(_view(p, nlev, ncol)), C.byref(_view(t, nlev, ncol)), C.byref(_view(td, nlev, ncol)),
                                  C.byref(pc), C.c_void_p(outs[0][1]), C.c_void_p(outs[1][1]), C.c_void_p(outs[2][1]), C.byref(so),
                                  kept.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nout), _stream(dev)))
    n = int(nout.value)
    arrs = [o[0].reshape((rows,) + hshape)[:n] for o in outs]
    parcel = {'pressure': par['parcel_pressure'].reshape(hshape), 'temperature': par['parcel_temperature'].reshape(hshape),
              'dewpoint': par['parcel_dewpoint'].reshape(hshape), 'index': par['parcel_index'].reshape(hshape)}
    return arrs[0], arrs[1], arrs[2], parcel, kept.astype(bool)


def from_most_unstable_parcel(pressure, temperature, dewpoint, depth=300):
    """pf.py:1517: (pressure, temperature, dewpoint) at and above each column's most-unstable parcel -- levels that no column
    keeps dropped, columns shifted onto their first kept level -- the parcel, and the mask of input levels that survived."""
    return _rebase(pressure, temperature, dewpoint, 'most_unstable', depth)


def mix_layer(pressure, temperature, dewpoint, depth=100):
    """pf.py:1604: the profiles with the lowest `depth` hPa replaced by the mixed parcel (row 0), the parcel, and the mask of
    input levels that survived."""
    p, t, td, parcel, kept = _rebase(pressure, temperature, dewpoint, 'mixed_layer', depth)
    parcel.pop('index')
    return p, t, td, parcel, kept


def interp1d(at, xp, fp):
    """pf.py:23 interp1d_numba = numpy.interp along the vertical: at (m, ...), xp / fp (n, ...) or (n,) shared by all
    columns; xp increasing along the vertical."""
    (ah,), dt, dev = _common(at)
    m, ncol, hshape = _vert_shape(ah)

    def pts(v):
        if _is_torch(v):
            h = _Arr(v.to(ah.t.device) if dev else v.cpu().numpy(), dtype=dt)
        else:
            v = np.asarray(v, dtype=dt)
            h = _Arr(torch.as_tensor(v).to(ah.t.device) if dev else v, dtype=dt)
        n = h.shape[0]
        cols = int(np.prod(h.shape[1:])) if len(h.shape) > 1 else 1
        assert cols in (1, ncol), 'xp / fp must have one column or one per column of `at`'
        return h, n, cols
    (xh, n, xc), (fh, n2, fc) = pts(xp), pts(fp)
    assert n == n2, 'xp and fp must have the same number of points'
    lib = L.init(_device_of(ah))
    out, optr = _alloc((m, ncol), dt, dev, ah)
    L.check(lib.xp_interp1d(C.byref(_view(ah, m, ncol)), C.byref(_view(xh, n, xc)), C.byref(_view(fh, n, fc)),
                            C.c_void_p(optr), _stream(dev)))
    return out.reshape((m,) + hshape)


def add_lcl_to_profile(profile, environment=None, interpolator='log'):
    """pf.py:858: profile = dict with pressure, temperature, virtual_temperature (nlev, ...) and lcl_pressure,
    lcl_temperature, lcl_virtual_temperature (...); environment = dict with pressure and variables.  Returns the profile
    with the LCL inserted as a level (nlev + 1 rows) and, per environment variable k, environment_k with the environment
    interpolated at the LCL inserted likewise (its virtual temperature recomputed from the interpolated temperature and
    dewpoint, pf.py:911-920)."""
    if interpolator not in ('linear', 'log'):
        raise AssertionError('interpolator must be linear or log')                      # pf.py:878
    level = {'pressure': profile['lcl_pressure'], 'temperature': profile['lcl_temperature'],
             'virtual_temperature': profile['lcl_virtual_temperature']}
    out = insert_level({k: profile[k] for k in level}, level, coords='pressure')
    for k in ('lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature'):
        out[k] = profile[k]
    if environment is not None:
        il = {k: interp_level(environment['pressure'], v, level['pressure'], log=(interpolator == 'log'))
              for k, v in environment.items()}
        il['pressure'] = level['pressure']
        if 'virtual_temperature' in il:
            il['virtual_temperature'] = virtual_temperature(il['temperature'],
                                                            mixing_ratio(il['temperature'], il['dewpoint'], il['pressure']))
        env = insert_level(environment, il, coords='pressure')
        for k in environment.keys():
            if k != 'pressure':
                out['environment_' + k] = env[k]
    return out


def freezing_level_height(temperature, height):
    """pf.py:2137."""
    return crossing_level(height, temperature, 273.15)


def wet_bulb_temperature_fast(temperature, dewpoint):
    """pf.py:364: the "1/3 rule" estimate -- plain array arithmetic, no kernel of its own."""
    return temperature - (1 / 3) * (temperature - dewpoint)


def melting_level_height(pressure, temperature, dewpoint, height, fast=True, moist=None):
    """pf.py:2160: freezing level of the wet-bulb temperature field; returns (height, wet bulb)."""
    wb = wet_bulb_temperature_fast(temperature, dewpoint) if fast else \
        wet_bulb_temperature(pressure, temperature, dewpoint, moist=moist)
    return crossing_level(height, wb, 273.15), wb


def isobar_temperature(pressure, temperature, isobar):
    """pf.py:2193."""
    return interp_level(pressure, temperature, isobar, log=True)


def lapse_rate(pressure, temperature, height, from_pressure=700, to_pressure=500):
    """pf.py:2102: (T_to - T_from) / (z_to - z_from) with z in km, all four by log-p interpolation."""
    t0 = interp_level(pressure, temperature, from_pressure, log=True)
    t1 = interp_level(pressure, temperature, to_pressure, log=True)
    z0 = interp_level(pressure, height, from_pressure, log=True) / 1000
    z1 = interp_level(pressure, height, to_pressure, log=True) / 1000
    return (t1 - t0) / (z1 - z0)


def deep_convective_index(pressure, temperature, dewpoint, lifted_index):
    """pf.py:1830 (Kunz 2009): T850 + Td850 [deg C] - LI."""
    t850 = interp_level(pressure, temperature, 850.0, log=True) - 273.15
    td850 = interp_level(pressure, dewpoint, 850.0, log=True) - 273.15
    return t850 + td850 - lifted_index


LIFTED_INDEX_VARS = ('pressure', 'temperature', 'environment_temperature')      # the profile rows lifted_index() reads


def lifted_index(profile):
    """pf.py:1722: environment minus parcel temperature at 500 hPa (log-p interpolation of the profile)."""
    env = interp_level(profile['pressure'], profile['environment_temperature'], 500.0, log=True)
    par = interp_level(profile['pressure'], profile['temperature'], 500.0, log=True)
    return env - par


# ---- product bundle (pf.py:1951-2100, 2216-2407): compositions of the calls above + array arithmetic ---------------
def _ns(x):
    """numpy-or-torch namespace shim for the few element-wise helpers the bundle needs."""
    if _is_torch(x):
        return torch
    return np


def _where(c, a, b):
    if _is_torch(c) or _is_torch(a) or _is_torch(b):
        dev = next(v.device for v in (c, a, b) if _is_torch(v))
        a = a if _is_torch(a) else torch.as_tensor(a, device=dev, dtype=torch.float64)
        b = b if _is_torch(b) else torch.as_tensor(b, device=dev, dtype=a.dtype)
        return torch.where(c, a, b.to(a.dtype))
    return np.where(c, a, b)


def _flat(hs):
    n = int(np.prod(hs[0].shape)) if hs[0].shape else 1
    assert all((int(np.prod(h.shape)) if h.shape else 1) == n for h in hs), 'per-point arguments must share a shape'
    return n


def _as_bool(x):
    return (x != 0)


def wind_shear(surface_wind_u, surface_wind_v, wind_u, wind_v, height, shear_height=6000):
    """pf.py:2216: wind at `shear_height` (linear interpolation in height) minus the surface wind (xp_wind_shear)."""
    (wu, wv, hh), dt, dev = _common(wind_u, wind_v, height)
    assert wu.shape == wv.shape == hh.shape, 'wind_u, wind_v, height must share a shape'
    nw, ncol, hshape = _vert_shape(wu)
    lib = L.init(_device_of(wu))
    su, sv = _per_col(surface_wind_u, ncol, dt, dev, wu), _per_col(surface_wind_v, ncol, dt, dev, wu)
    outs = [_alloc((ncol,), dt, dev, wu) for _ in range(3)]
    pos, pptr = _alloc((ncol,), np.int32, dev, wu)
    L.check(lib.xp_wind_shear(C.byref(_view(wu, nw, ncol)), C.byref(_view(wv, nw, ncol)), C.byref(_view(hh, nw, ncol)),
                              C.c_void_p(su.ptr), C.c_void_p(sv.ptr), C.c_double(float(shear_height)), C.c_void_p(outs[0][1]),
                              C.c_void_p(outs[1][1]), C.c_void_p(outs[2][1]), C.c_void_p(pptr), _stream(dev)))
    return {'shear_u': outs[0][0].reshape(hshape), 'shear_v': outs[1][0].reshape(hshape),
            'shear_magnitude': outs[2][0].reshape(hshape), 'positive_shear': _as_bool(pos).reshape(hshape)}


def significant_hail_parameter(mucape, mixing_ratio, lapse, temp_500, shear, flh):
    """pf.py:2261 (SPC SHIP) with the reference's validity windows (xp_significant_hail_parameter)."""
    hs, dt, dev = _common(mucape, mixing_ratio, lapse, temp_500, shear, flh)
    n = _flat(hs)
    lib = L.init(_device_of(hs[0]))
    out, optr = _alloc((n,), dt, dev, hs[0])
    L.check(lib.xp_significant_hail_parameter(C.c_int64(n), C.c_int32(hs[0].xp_dtype), C.c_int32(hs[0].mem),
                                              *[C.c_void_p(h.ptr) for h in hs], C.c_void_p(optr), _stream(dev)))
    return out.reshape(hs[0].shape)


def conv_properties(dat, ignore_nans=False, moist=None):
    """pf.py:1951: the reference's convective-property bundle for a grid, ONE library call (xp_conv_properties): the
    q -> dewpoint step, the NaN mask, the fixed-level interpolations and the freezing / melting levels are one pass over
    the four grids, the three parcels' CAPE / CIN / lifted index three more, the rest one per-point kernel -- no array
    arithmetic on this side.  `dat`: mapping with pressure [hPa], temperature [K], specific_humidity [kg/kg],
    height_asl [m] (nlev, ...), wind_u, wind_v, wind_height_above_surface (nwind, ...), surface_wind_u, surface_wind_v
    (...): NumPy arrays (staged through the library) or torch CUDA tensors (in place).  Returns a dict of per-column
    arrays with the reference's variable names (positive_shear: bool)."""
    order = L.CONV_IN_VIEWS + ('surface_wind_u', 'surface_wind_v')
    hs, dt, dev = _common(*[dat[k] for k in order])
    h = dict(zip(order, hs))
    p = h['pressure']
    assert all(h[k].shape == p.shape for k in ('temperature', 'specific_humidity', 'height_asl')), 'pressure, temperature, specific_humidity, height_asl must share a shape'
    nlev, ncol, hshape = _vert_shape(p)
    wu = h['wind_u']
    assert wu.shape == h['wind_v'].shape == h['wind_height_above_surface'].shape and tuple(wu.shape[1:]) == hshape, 'wind arrays must be (nwind, ...) over the same points'
    nwind = wu.shape[0]
    lib = L.init(_device_of(p))
    views = {k: _view(h[k], nlev if k in L.CONV_IN_VIEWS[:4] else nwind, ncol) for k in L.CONV_IN_VIEWS}
    ci = L.ConvIn(*[C.pointer(views[k]) for k in L.CONV_IN_VIEWS], h['surface_wind_u'].ptr, h['surface_wind_v'].ptr)
    assert int(np.prod(h['surface_wind_u'].shape)) == ncol and int(np.prod(h['surface_wind_v'].shape)) == ncol
    co, out = L.ConvOut(), {}
    for k in L.CONV_OUT:
        arr, ptr = _alloc((ncol,), np.int32 if k == 'positive_shear' else dt, dev, p)
        setattr(co, k, ptr)
        out[k] = arr
    o = _opts(moist=moist or _DEFAULT['moist'])
    L.check(lib.xp_conv_properties(C.byref(ci), C.byref(o), C.c_int32(int(bool(ignore_nans))), C.byref(co), _stream(dev)))
    res = {k: v.reshape(hshape) for k, v in out.items()}
    res['positive_shear'] = res['positive_shear'] != 0
    return res


def conv_properties_composed(dat, ignore_nans=False, moist=None):
    """The same bundle as a composition of the stand-alone calls plus array arithmetic (what conv_properties() was before
    xp_conv_properties existed): kept as the cross-check of the fused call (tests/test_gpu_indices.py)."""
    host_in = not any(_is_torch(v) for v in dat.values())
    if host_in and torch is not None and torch.cuda.is_available():
        # one upload; the ~25 kernel launches of the bundle then work on device-resident data
        dat = {k: torch.as_tensor(np.ascontiguousarray(np.asarray(v, dtype=np.float64))).cuda() for k, v in dat.items()}
    p, t, q = dat['pressure'], dat['temperature'], dat['specific_humidity']
    td = dewpoint_from_specific_humidity(p, t, q)
    xp = _ns(td)
    to = (lambda a: torch.as_tensor(np.asarray(a), device=td.device) if not _is_torch(a) else a) if _is_torch(td) else np.asarray
    p, t, q, z = to(p), to(t), to(q), to(dat['height_asl'])
    valid = ~(xp.isnan(td).any(0) | xp.isnan(p).any(0) | xp.isnan(t).any(0) | xp.isnan(q).any(0))
    out = {}
    # (lifted_index_at: pf.py:1722 on the lifted profile, in the same pass, instead of writing the profile and interpolating it)
    mu = cape_cin_columns(p, t, td, parcel='most_unstable', depth=250, lifted_index_at=500.0, moist=moist)
    out['mu_cape'], out['mu_cin'] = mu['cape'], mu['cin']
    e = 6.112 * xp.exp(17.67 * (mu['parcel_dewpoint'] - 273.15) / (mu['parcel_dewpoint'] - 29.65))
    w = L_EPS * e / (mu['parcel_pressure'] - e)                       # specific_humidity_from_dewpoint -> mixing ratio
    qs = w / (1.0 + w)
    out['mu_mixing_ratio'] = qs / (1.0 - qs)
    out['mu_lifted_index'] = mu['lifted_index']
    for depth in (100, 50):
        ml = cape_cin_columns(p, t, td, parcel='mixed_layer', depth=depth, lifted_index_at=500.0, moist=moist)
        out[f'mixed_{depth}_cape'], out[f'mixed_{depth}_cin'] = ml['cape'], ml['cin']
        out[f'mixed_{depth}_lifted_index'] = ml['lifted_index']
    # temperature, dewpoint and height at 850 / 700 / 500 hPa in ONE pass over the column: what deep_convective_index
    # (pf.py:1830), lapse_rate (pf.py:2102) and isobar_temperature (pf.py:2193) interpolate one launch at a time
    (t850, t700, t500), (td850, _, _), (_, z700, z500) = interp_levels(p, [t, td, z], [850.0, 700.0, 500.0], log=True)
    for pre in ('mu', 'mixed_100', 'mixed_50'):
        out[pre + '_dci'] = (t850 - 273.15) + (td850 - 273.15) - out[pre + '_lifted_index']
    out['lapse_rate_700_500'] = (t500 - t700) / (z500 / 1000 - z700 / 1000)
    out['temp_500'] = t500
    out['freezing_level'] = freezing_level_height(t, z)
    out['melting_level'], _ = melting_level_height(p, t, td, z)
    out.update(wind_shear(to(dat['surface_wind_u']), to(dat['surface_wind_v']), to(dat['wind_u']), to(dat['wind_v']),
                          to(dat['wind_height_above_surface'])))
    if not ignore_nans:
        for k in out:
            if k != 'positive_shear':
                out[k] = _where(valid, out[k], float('nan'))
            else:
                out[k] = out[k] & valid          # xarray's where() turns a masked boolean into NaN; here: False
    if host_in:
        out = {k: (v.cpu().numpy() if _is_torch(v) else v) for k, v in out.items()}
    return out


def min_conv_properties(dat, moist=None):
    """pf.py:1873: the minimal bundle -- 100 hPa mixed-layer CAPE / CIN and lifted index, 700-500 hPa lapse rate, 500 hPa
    temperature, freezing and melting level, 0-6 km shear.  Same input mapping as conv_properties(); no NaN blanking
    (the reference has none here)."""
    host_in = not any(_is_torch(v) for v in dat.values())
    if host_in and torch is not None and torch.cuda.is_available():
        dat = {k: torch.as_tensor(np.ascontiguousarray(np.asarray(v, dtype=np.float64))).cuda() for k, v in dat.items()}
    p, t, z = dat['pressure'], dat['temperature'], dat['height_asl']
    td = dewpoint_from_specific_humidity(p, t, dat['specific_humidity'])
    ml = cape_cin_columns(p, t, td, parcel='mixed_layer', depth=100, lifted_index_at=500.0, moist=moist)
    (t700, t500), (z700, z500) = interp_levels(p, [t, z], [700.0, 500.0], log=True)      # pf.py:2102, 2193 in one pass
    out = {'mixed_100_cape': ml['cape'], 'mixed_100_cin': ml['cin'], 'mixed_100_lifted_index': ml['lifted_index'],
           'lapse_rate_700_500': (t500 - t700) / (z500 / 1000 - z700 / 1000), 'temp_500': t500,
           'freezing_level': freezing_level_height(t, z), 'melting_level': melting_level_height(p, t, td, z)[0]}
    out.update(wind_shear(dat['surface_wind_u'], dat['surface_wind_v'], dat['wind_u'], dat['wind_v'],
                          dat['wind_height_above_surface']))
    if host_in:
        out = {k: (v.cpu().numpy() if _is_torch(v) else v) for k, v in out.items()}
    return out


def storm_proxies(dat):
    """pf.py:2323: hail / storm proxies (booleans) and SHIP from the output of conv_properties(), one per-point kernel
    (xp_storm_proxies)."""
    hs, dt, dev = _common(*[dat[k] for k in L.PROXIES_IN])
    n = _flat(hs)
    shape = hs[0].shape
    lib = L.init(_device_of(hs[0]))
    ps = dat['positive_shear']
    if _is_torch(ps):
        ps = ps.to(torch.int32)
        ps = ps.to(hs[0].t.device) if dev else ps.cpu().numpy()
    else:
        ps = np.asarray(ps).astype(np.int32)
        if dev:
            ps = torch.as_tensor(ps).to(hs[0].t.device)
    ps = ps.reshape(-1).contiguous() if _is_torch(ps) else np.ascontiguousarray(ps.reshape(-1))
    assert int(np.prod(ps.shape)) == n, 'positive_shear does not match the other arrays'
    pin = L.ProxiesIn(*[h.ptr for h in hs], ps.data_ptr() if _is_torch(ps) else ps.ctypes.data)
    flags = [_alloc((n,), np.int32, dev, hs[0]) for _ in L.PROXIES_OUT]
    ship, sptr = _alloc((n,), dt, dev, hs[0])
    pout = L.ProxiesOut(*[f[1] for f in flags], sptr)
    L.check(lib.xp_storm_proxies(C.c_int64(n), C.c_int32(hs[0].xp_dtype), C.c_int32(hs[0].mem), C.byref(pin), C.byref(pout),
                                 _stream(dev)))
    out = {k: _as_bool(f[0]).reshape(shape) for k, f in zip(L.PROXIES_OUT, flags)}
    out['ship'] = ship.reshape(shape)
    # the reference's order of variables (pf.py:2395-2405): proxies, SHIP, the SHIP proxy
    return {**{k: out[k] for k in L.PROXIES_OUT[:8]}, 'ship': out['ship'], 'proxy_SHIP_0.1': out['proxy_SHIP_0.1']}


def family_table():
    """The adiabat-family coefficient table of moist='family' as a (n_coef_rows, n_label_pieces) float64 array: rows are
    (x-piece, power of z, power of s), C-order (xp_family_table)."""
    lib = L.init()
    n1, n2 = C.c_int64(), C.c_int64()
    L.check(lib.xp_family_table(None, C.byref(n1), C.byref(n2)))
    out = np.empty((n1.value, n2.value), dtype=np.float64)
    L.check(lib.xp_family_table(out.ctypes.data_as(C.c_void_p), None, None))
    return out


def set_family_table(table):
    """Replace the adiabat-family table (xp_set_family_table)."""
    t = np.ascontiguousarray(table, dtype=np.float64)
    t = t.reshape(-1, t.shape[-1])
    L.check(L.init().xp_set_family_table(t.ctypes.data_as(C.c_void_p), C.c_int64(t.shape[0]), C.c_int64(t.shape[1])))
