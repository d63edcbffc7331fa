"""ctypes binding of libxparcel.so (include/xparcel.h).  No CPU fallback: if the library is missing
or no MI355X is visible, calls raise."""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XPARCEL_LIB') or os.path.join(_HERE, 'lib', 'libxparcel.so')   # env override: A/B builds
SRC_DIR = os.path.join(_HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include', 'xparcel.h')

XP_F32, XP_F64 = 0, 1
XP_MEM_HOST, XP_MEM_DEVICE = 0, 1
PARCEL = {'surface': 0, 'most_unstable': 1, 'mixed_layer': 2, 'explicit': 3}
MOIST = {'exact': 0, 'table': 1, 'family': 2}
LCL_INTERP = {'linear': 0, 'log': 1}
HUMIDITY = {'dewpoint': 0, 'specific': 1}
OPT_FUSE_PARCELS = 1
ST_TOP_NAN, ST_LCL_NOT_CONVERGED, ST_NAN_PRESSURE, ST_BAD_PRESSURE = 1, 2, 4, 8

# every symbol include/xparcel.h declares
SYMBOLS = ('xp_version', 'xp_init', 'xp_set_tables', 'xp_tables_loaded', 'xp_family_table', 'xp_set_family_table', 'xp_cape_cin', 'xp_cape_cin_multi', 'xp_lcl', 'xp_dry_lapse',
           'xp_moist_lapse', 'xp_parcel_profile', 'xp_lfc_el', 'xp_cape_cin_base', 'xp_select_parcel',
           'xp_mixed_layer', 'xp_wet_bulb_temperature', 'xp_interp_level', 'xp_interp_levels', 'xp_dewpoint_from_specific_humidity',
           'xp_crossing_level', 'xp_mixing_ratio', 'xp_conv_properties', 'xp_insert_level', 'xp_find_intersections', 'xp_trapz',
           'xp_trap_around_zeros', 'xp_bound_pressure', 'xp_get_layer', 'xp_shift_out_nans', 'xp_rebase_profile', 'xp_interp1d',
           'xp_wind_shear', 'xp_significant_hail_parameter', 'xp_storm_proxies',
           'xp_last_error')


class View(C.Structure):
    _fields_ = [('data', C.c_void_p), ('dtype', C.c_int32), ('mem', C.c_int32), ('nlev', C.c_int64),
                ('ncol', C.c_int64), ('lev_stride', C.c_int64), ('col_stride', C.c_int64)]


class Parcel(C.Structure):
    _fields_ = [('mode', C.c_int32), ('reserved', C.c_int32), ('depth', C.c_double), ('pressure', C.c_void_p),
                ('temperature', C.c_void_p), ('dewpoint', C.c_void_p)]


class Opts(C.Structure):
    _fields_ = [('virtual_temperature_correction', C.c_int32), ('lcl_interp', C.c_int32),
                ('pos_cape_neg_cin', C.c_int32), ('post_zero_cin', C.c_int32), ('moist_mode', C.c_int32),
                ('compute', C.c_int32), ('humidity', C.c_int32), ('flags', C.c_int32)]


SCALAR_F = ('cape', 'cin', 'lcl_pressure', 'lcl_temperature', 'lcl_virtual_temperature', 'lfc_pressure',
            'lfc_temperature', 'el_pressure', 'el_temperature')
SCALAR_I = ('lfc_index', 'el_index', 'status', 'parcel_index')
SCALAR_P = ('parcel_pressure', 'parcel_temperature', 'parcel_dewpoint')


class ScalarsOut(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in SCALAR_F] + [(k, C.c_void_p) for k in SCALAR_I] +
                [(k, C.c_void_p) for k in SCALAR_P] + [('dtype', C.c_int32), ('mem', C.c_int32)])


PROFILE_VARS = ('pressure', 'temperature', 'virtual_temperature', 'environment_temperature',
                'environment_virtual_temperature', 'environment_dewpoint')


class ProfileOut(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in PROFILE_VARS] +
                [('dtype', C.c_int32), ('mem', C.c_int32), ('nlev_out', C.c_int64), ('lev_stride', C.c_int64),
                 ('col_stride', C.c_int64), ('lifted_index', C.c_void_p), ('lifted_index_pressure', C.c_double)])


CONV_IN_VIEWS = ('pressure', 'temperature', 'specific_humidity', 'height_asl', 'wind_u', 'wind_v', 'wind_height_above_surface')
CONV_OUT = ('mu_cape', 'mu_cin', 'mu_mixing_ratio', 'mu_lifted_index', 'mu_dci', 'mixed_100_cape', 'mixed_100_cin',
            'mixed_100_lifted_index', 'mixed_100_dci', 'mixed_50_cape', 'mixed_50_cin', 'mixed_50_lifted_index', 'mixed_50_dci',
            'lapse_rate_700_500', 'temp_500', 'freezing_level', 'melting_level', 'shear_u', 'shear_v', 'shear_magnitude',
            'positive_shear')


class ConvIn(C.Structure):
    _fields_ = [(k, C.POINTER(View)) for k in CONV_IN_VIEWS] + [('surface_wind_u', C.c_void_p), ('surface_wind_v', C.c_void_p)]


class ConvOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in CONV_OUT]


PROXIES_IN = ('mu_cape', 'mu_mixing_ratio', 'mixed_100_cape', 'mixed_100_cin', 'mixed_100_lifted_index', 'mixed_100_dci',
              'mixed_50_cape', 'mixed_50_cin', 'lapse_rate_700_500', 'temp_500', 'freezing_level', 'shear_magnitude')
PROXIES_OUT = ('proxy_Craven2004', 'proxy_Kunz2007', 'proxy_Trapp2007', 'proxy_Marsh2009', 'proxy_Allen2011', 'proxy_Allen2014',
               'proxy_Eccel2012', 'proxy_Mohr2013', 'proxy_SHIP_0.1')


class ProxiesIn(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in PROXIES_IN] + [('positive_shear', C.c_void_p)]


class ProxiesOut(C.Structure):
    _fields_ = [('f%d' % i, C.c_void_p) for i in range(9)] + [('ship', C.c_void_p)]


class Tables(C.Structure):
    _fields_ = [('n_pressure', C.c_int64), ('n_temperature', C.c_int64), ('n_adiabat', C.c_int64),
                ('p_max', C.c_double), ('p_step', C.c_double), ('t_min', C.c_double), ('t_step', C.c_double),
                ('index', C.c_void_p), ('adiabats', C.c_void_p)]


class XParcelError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libxparcel error {code}: {msg}')
        self.code = code


HIPCC_FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC'] + os.environ.get('XP_EXTRA_DEFS', '').split()   # (A/B builds)
# translation units: the ABI + small kernels, and k_cape_cin once per (dtype, moist mode) -- see csrc/xp_cape_tu.hip
# Per-moist-mode compile flags of the k_cape_cin translation units (XP_TU_FLAGS_<mode> overrides them for A/B builds).
# Mode 2 (adiabat family): ONE 1024-thread workgroup per CU -- the family coefficient table (46.7 KB) is staged into LDS
# next to the e_s / ln tables and the per-thread scan slots (157.5 of the CU's 160 KB) -- which caps the kernel at 128
# VGPRs; -disable-machine-licm keeps the compiler from hoisting the materialisation of ~40 fp64 constants out of the
# level loop into registers it then has to spill (128 VGPRs + 200 B of scratch with it, no spills in the loop without;
# the RK4 kernels lose ~2 % to the flag and do not get it).
TU_FLAGS = {0: [], 1: [], 2: ['-DXP_CAPE_THREADS=1024', '-mllvm', '-disable-machine-licm']}
for _m in list(TU_FLAGS):
    if os.environ.get(f'XP_TU_FLAGS_{_m}') is not None:
        TU_FLAGS[_m] = os.environ[f'XP_TU_FLAGS_{_m}'].split()
# The fused several-parcels kernel (csrc/xp_multi.hpp), one unit per (dtype, number of parcels): workgroup size and LDS
# slot fields per chain such that 58.6 KB of tables + NP x fields x threads x 8 B fit the CU's 160 KB.
MULTI_FLAGS = {2: ['-DXP_CAPE_THREADS=512', '-DXP_SLOT_FIELDS=12', '-mllvm', '-disable-machine-licm']}
for _m in list(MULTI_FLAGS):
    if os.environ.get(f'XP_MULTI_FLAGS_{_m}') is not None:
        MULTI_FLAGS[_m] = os.environ[f'XP_MULTI_FLAGS_{_m}'].split()
UNITS = [('xparcel', 'xparcel.hip', [])] + [
    (f'cape_{t[0]}{m}', 'xp_cape_tu.hip', [f'-DXP_TU_T={t}', f'-DXP_TU_MODE={m}'] + TU_FLAGS[m])
    for t in ('double', 'float') for m in (0, 1, 2)] + [
    (f'multi_{t[0]}{n}', 'xp_multi_tu.hip', [f'-DXP_TU_T={t}', f'-DXP_MULTI_NP={n}'] + MULTI_FLAGS[n])
    for t in ('double', 'float') for n in sorted(MULTI_FLAGS)]


def build(force=False, verbose=False, jobs=None):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU): the translation units in
    parallel (hipcc -c), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(SRC_DIR, f) for f in sorted(os.listdir(SRC_DIR))] + [INCLUDE]
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    libdir = os.path.dirname(LIB_PATH)
    objdir = os.path.join(libdir, 'obj%d' % os.getpid())
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

    def compile_unit(u):
        name, src, defs = u
        obj = os.path.join(objdir, name + '.o')
        cmd = [hipcc] + HIPCC_FLAGS + defs + ['-c', os.path.join(SRC_DIR, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    try:
        with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_unit, UNITS))
        tmp = LIB_PATH + '.tmp%d' % os.getpid()
        cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', tmp] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, LIB_PATH)       # atomic: a concurrent loader never sees a half-written library
    finally:
        import shutil
        shutil.rmtree(objdir, ignore_errors=True)
    return LIB_PATH


def csrc_sha():
    """Fingerprint of the kernel sources (csrc/ + the ABI header): profiles taken on the GPU box record it, and bench.py
    only quotes a profile's counters next to a timing when the fingerprints match."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(SRC_DIR)) + [INCLUDE]:
        path = f if os.path.isabs(f) else os.path.join(SRC_DIR, f)
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None
_lock = threading.Lock()
_inited_device = None


def load():
    """dlopen the library (no GPU needed for this)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                                  '(there is no CPU fallback)')
            # torch ships its own HIP runtime: when torch is going to be used in this process it has to be the first to
            # load one, or the library's hipGetDeviceCount() later reports "no ROCm-capable device" (two runtimes in one
            # process).  torch is plumbing here (device memory, streams), so importing it is not a product dependency.
            try:
                import torch  # noqa: F401
            except Exception:
                pass
            _lib = C.CDLL(LIB_PATH)
            _lib.xp_last_error.restype = C.c_char_p
            for s in SYMBOLS:
                getattr(_lib, s)
    return _lib


def check(rc):
    if rc != 0:
        raise XParcelError(rc, load().xp_last_error().decode())


def init(device=None):
    """xp_init on the given (or torch-current, or 0) device."""
    global _inited_device
    lib = load()
    if device is None:
        device = _inited_device if _inited_device is not None else int(os.environ.get('LOCAL_RANK', '0'))
    if _inited_device != device:
        check(lib.xp_init(int(device)))
        _inited_device = device
    return lib
