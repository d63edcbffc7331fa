/*
 * xparcel.h -- C ABI of libxparcel (MI355X / gfx950), the drop-in boundary for the
 * parcel-lifting hot path of traupach/xarray_parcel.
 *
 * The reference has no FFI layer of its own: its boundary is the set of Python
 * functions in modules/parcel_functions.py ("pf.py").  Each entry point below names
 * the reference function it replaces (file:line); the host-side mirror in
 * xarray_parcel_amd/parcel_functions.py keeps the reference's signatures and calls
 * these through ctypes (INTEGRATION.md shows the binding a maintainer of the
 * reference would add).
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer; the library never
 *     allocates caller-visible memory and never frees caller memory;
 *   - arrays are (nlev, ncol): level 0 = surface, pressure strictly decreasing
 *     upwards (README.md:9, pf.py:2319-2320), hPa / K / K, NaN = missing; element
 *     (k, c) of a view lives at data[k*lev_stride + c*col_stride] (strides in
 *     elements).  col_stride == 1 (the (lev, y, x) C-order layout) is the coalesced
 *     fast path; anything else is correct but slower;
 *   - dtype is XP_F32 or XP_F64 for data in memory; arithmetic is fp64;
 *   - mem says where a buffer lives: XP_MEM_DEVICE pointers are used in place,
 *     XP_MEM_HOST buffers are staged through internal device scratch (PCIe time
 *     is then part of the call);
 *   - every function returns 0 on success or a negative XP_E* code; the message is
 *     available from xp_last_error() (thread-local).  The reference's data-dependent
 *     asserts (pf.py:131, 1149, 1158) become per-column status bits, not aborts;
 *   - one device per process (the deployment model is one process per GPU): xp_init(device)
 *     binds the library's state -- lookup tables, the e_s / ln / adiabat-family tables -- to that
 *     device, and every call runs there whatever the calling thread's current device is (which
 *     is restored on return).  Calls from several host threads are safe against each other;
 *     replacing tables (xp_set_tables, xp_set_family_table, xp_init on another device) waits for
 *     the device to drain first and must not race with other calls;
 *   - work is enqueued on the hipStream_t passed as `stream` (NULL = the default stream), which
 *     must belong to the library's device.  Device-resident calls return without synchronising
 *     (an asynchronous kernel fault surfaces at the caller's next synchronisation); calls with
 *     host buffers synchronise the stream before returning.
 */
#ifndef XPARCEL_H
#define XPARCEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XP_VERSION 100 /* 0.1.0 */

enum { XP_F32 = 0, XP_F64 = 1 };
enum { XP_MEM_HOST = 0, XP_MEM_DEVICE = 1 };

/* which parcel is lifted (pf.py:1477 / 1557 / 1651 / 1394) */
enum { XP_PARCEL_SURFACE = 0, XP_PARCEL_MOST_UNSTABLE = 1, XP_PARCEL_MIXED_LAYER = 2, XP_PARCEL_EXPLICIT = 3 };

/* moist adiabat: XP_MOIST_EXACT integrates MetPy's pseudo-adiabat ODE (what the reference's
   KATs are run with, unit_tests.py:114-140) by RK4 in ln p with steps <= 0.1;
   XP_MOIST_TABLE emulates the reference's lookup tables (pf.py:525-607) against tables given
   to xp_set_tables(). */
enum { XP_MOIST_EXACT = 0, XP_MOIST_TABLE = 1, XP_MOIST_FAMILY = 2 };
/* XP_HUM_SPECIFIC: the moisture view holds specific humidity [kg/kg] (what the reference's data files provide) and is
   converted to dewpoint on load with the xp_dewpoint_from_specific_humidity chain (parcel_test.py:262-266), saving the
   separate pass and the (nlev, ncol) dewpoint array.  Explicit parcels (xp_parcel.dewpoint) stay dewpoints. */
enum { XP_HUM_DEWPOINT = 0, XP_HUM_SPECIFIC = 1 };
/* XP_MOIST_FAMILY: the same pseudo-adiabat ODE, served from a piecewise-polynomial table of its solutions: the parcel's
   VIRTUAL temperature along the adiabat, Tv(ln p ; psi) = T (1 + 0.608 w_s(p, T)) (what pf.py:760-775 feed the CAPE / CIN
   integration), psi = the adiabat's temperature at 1000 hPa (8 pieces of 0.5 in ln p from 1100 hPa to ~20 hPa x 9 pieces
   in psi from 215 to 312 K, degree 8 x 8, 46.7 KB, built at xp_init; specification: oracle/family.py).  Within 1e-6 K
   of the ODE (the RK4 stepper of XP_MOIST_EXACT: 2e-5 K; MetPy's LSODA: 4e-5 ... 4e-4 K), a level costs one Horner
   evaluation instead of an RK4 step, and all of the reference's known-answer tests pass in this mode too.  The parcel
   TEMPERATURE, where asked for (profile output, virtual_temperature_correction off), is the T with that virtual
   temperature at that pressure (Newton on the reference's own formula).  Above the table's top the adiabat continues dry.
   Columns whose label or LCL leave the table (psi outside 215.05 ... 311.95 K, p_lcl outside ~20 ... 1100 hPa) are
   transparently redone with XP_MOIST_EXACT.  Honoured by xp_cape_cin; the component entry points treat it as
   XP_MOIST_EXACT. */
enum { XP_LCL_INTERP_LINEAR = 0, XP_LCL_INTERP_LOG = 1 };
/* xp_opts.flags.  XP_OPT_FUSE_PARCELS: xp_cape_cin_multi lifts its parcels in ONE pass over the grid where it can (see
   there); same results bit for bit, measured SLOWER than one pass per parcel on MI355X, so off by default. */
enum { XP_OPT_FUSE_PARCELS = 1 };

/* error codes */
enum {
    XP_OK = 0,
    XP_E_ARG = -1,          /* null / inconsistent arguments (shape, dtype, mem mismatch)       */
    XP_E_NOT_INIT = -2,     /* xp_init not called                                              */
    XP_E_NO_TABLES = -3,    /* table mode without tables: 'Call load_moist_adiabat_lookups first' (pf.py:60) */
    XP_E_INTERP = -4,       /* 'interpolator must be linear or log' (pf.py:878)                */
    XP_E_HIP = -5,          /* HIP runtime error (message in xp_last_error)                    */
    XP_E_NO_DEVICE = -6     /* no usable gfx950 device                                         */
};

/* per-column status bits (xp_scalars_out.status) */
enum {
    XP_ST_TOP_NAN = 1,          /* 'Top temperature is NaN' condition of pf.py:1149 */
    XP_ST_LCL_NOT_CONVERGED = 2,/* LCL fixed point hit 50 iterations (MetPy raises)  */
    XP_ST_BAD_PRESSURE = 8,     /* a pressure higher than the level below it: outside the input contract (README.md:9,
                                   pf.py:2319-2320); the column's other outputs are unspecified.  (A pressure <= 0 has no
                                   logarithm: it is not tested for as such, but in practice trips this test too.) */
    XP_ST_NAN_PRESSURE = 4      /* a NaN pressure below the LCL.  The level is treated as MISSING -- exactly as if its
                                   temperature and dewpoint were NaN too: the two intervals that touch it drop out of
                                   every sum and the LCL bracket skips it -- not as the reference's insert_level does
                                   (pf.py:962-966 puts a copy of the LCL into the NaN slot and integrates over the
                                   out-of-order profile).  Contract: tests/test_gpu_parity.py::test_nan_pressure_levels */
};

typedef struct {
    const void *data;
    int32_t dtype;      /* XP_F32 | XP_F64 */
    int32_t mem;        /* XP_MEM_HOST | XP_MEM_DEVICE */
    int64_t nlev, ncol;
    int64_t lev_stride, col_stride; /* in elements.  xp_cape_cin is fastest when its three views share their strides
                                       (non-negative, < 4 GiB per level row): anything else is copied to dense scratch first */
} xp_view;

typedef struct {
    int32_t mode;       /* XP_PARCEL_* */
    int32_t reserved;
    double depth;       /* hPa: MU search depth (default 300, pf.py:1558) / ML mixing depth (100, pf.py:1652) */
    /* XP_PARCEL_EXPLICIT only: per-column parcel, ncol elements each, same dtype/mem as the views */
    const void *pressure, *temperature, *dewpoint;
} xp_parcel;

typedef struct {
    int32_t virtual_temperature_correction; /* default 1 (pf.py:1396) */
    int32_t lcl_interp;                     /* XP_LCL_INTERP_*; default log (pf.py:1396) */
    int32_t pos_cape_neg_cin;               /* default 1 (pf.py:1293) */
    int32_t post_zero_cin;                  /* default 0 (pf.py:1293) */
    int32_t moist_mode;                     /* XP_MOIST_* */
    int32_t compute;                        /* arithmetic type: XP_F64 (the only one implemented; XP_F32 is rejected with XP_E_ARG) */
    int32_t humidity;                       /* XP_HUM_*: what the `dewpoint` view of xp_cape_cin holds */
    int32_t flags;                          /* XP_OPT_* bits; 0 by default */
} xp_opts;

/* Per-column outputs; every pointer is nullable (not written when NULL).  Floating outputs have
   `dtype`, all buffers live in `mem`, ncol elements each.
   lfc_index / el_index: index i of the interval (levels i, i+1 of the LCL-augmented profile)
   that holds the chosen crossing; -1 = none; lfc_index -2 = LFC replaced by the LCL (pf.py:1161-1185).
   parcel_index: source level of the lifted parcel (0 for surface, MU level for most-unstable, -1 otherwise).
   Cost: with default options and lfc_temperature, el_temperature, lfc_index, el_index and status all NULL the
   pass runs the CAPE/CIN-only kernels (3-5 % faster); any of the five selects the all-outputs kernels.  Values
   of the arrays that are written do not depend on which kernel ran. */
typedef struct {
    void *cape, *cin;                                   /* J/kg (pf.py:1361-1385) */
    void *lcl_pressure, *lcl_temperature, *lcl_virtual_temperature; /* pf.py:609-682 */
    void *lfc_pressure, *lfc_temperature, *el_pressure, *el_temperature; /* pf.py:1066-1198 */
    int32_t *lfc_index, *el_index, *status, *parcel_index;
    void *parcel_pressure, *parcel_temperature, *parcel_dewpoint; /* the lifted parcel (MU: pf.py:133, ML: pf.py:268-287) */
    int32_t dtype, mem;
} xp_scalars_out;

/* Optional lifted profile with the LCL inserted as an extra level (pf.py:806-931): six
   (nlev_out, ncol) arrays, nlev_out >= nlev + 1.  For MU / ML parcels the profile is re-based
   (levels below the parcel removed, pf.py:1551-1553, 1636-1644) and padded with NaN on top.  Any of the six
   pointers may be NULL: that array is not written (lifted_index, pf.py:1722, reads three of them). */
typedef struct {
    void *pressure, *temperature, *virtual_temperature;                 /* parcel */
    void *environment_temperature, *environment_virtual_temperature, *environment_dewpoint;
    int32_t dtype, mem;
    int64_t nlev_out, lev_stride, col_stride;
    /* lifted_index (pf.py:1722): environment minus parcel temperature of THIS profile at `lifted_index_pressure` hPa
       (the reference: 500), by the log_interp rule of pf.py:1813 applied to the profile's rows -- ncol values of the
       profile's dtype / mem, written in the same pass; NULL = not wanted.  With all six arrays NULL the pass costs
       little more than CAPE / CIN alone -- in family mode, with the default options and a CAPE/CIN-only xp_scalars_out
       (see there), 10 % more: the parcel's plain temperature is then derived from the table's virtual temperature only at
       the two nodes around the level (any other request in family mode derives it at every node: + 65 %). */
    void *lifted_index;
    double lifted_index_pressure;
} xp_profile_out;

/* reference-format moist-adiabat lookup tables (pf.py:447-523), host memory, copied to the device */
typedef struct {
    int64_t n_pressure, n_temperature, n_adiabat;
    double p_max, p_step;       /* index-grid pressures  p_max - i*p_step   (1100 ... 2.5, pf.py:447-448) */
    double t_min, t_step;       /* index-grid temperatures t_min + j*t_step (173 ... 315.98, pf.py:449-450) */
    const uint16_t *index;      /* [n_pressure][n_temperature] adiabat number, 0 = NaN */
    const float *adiabats;      /* [n_adiabat][n_pressure], pressure ASCENDING (pf.py:54) */
} xp_tables;

int xp_version(void);

/* Select the device and create the library state; idempotent.  Replaces the module-global set-up of
   pf.py:18-21.  device = HIP device ordinal. */
int xp_init(int device);

/* pf.py:39-61 load_moist_adiabat_lookups: hand over tables (generated by
   xarray_parcel_amd.adiabat_tables or loaded from its cache file). */
int xp_set_tables(const xp_tables *tables);
int xp_tables_loaded(void);

/* The coefficient table of XP_MOIST_FAMILY as [n_lnp][n_label] doubles: n_lnp = 648 rows (x-piece, power of z, power of s,
   C-order), n_label = 9 psi-pieces (no reference counterpart: the reference's tables are the XP_MOIST_TABLE ones).  Read
   it back (out may be NULL to query the shape) or replace it, e.g. with the oracle's independently built copy in the
   parity tests. */
int xp_family_table(double *out, int64_t *n_lnp, int64_t *n_label);
int xp_set_family_table(const double *table, int64_t n_lnp, int64_t n_label);

/* pf.py:1394-1475 cape_cin and its three drivers: surface_based_cape_cin (pf.py:1477),
   most_unstable_cape_cin (pf.py:1557), mixed_layer_cape_cin (pf.py:1651); with `profile` non-NULL
   also parcel_profile_with_lcl (pf.py:806) + lfc_el (pf.py:1066) in the same pass. */
int xp_cape_cin(const xp_view *pressure, const xp_view *temperature, const xp_view *dewpoint,
                const xp_parcel *parcel, const xp_opts *opts,
                xp_scalars_out *scalars, xp_profile_out *profile, void *stream);

/* Several parcels of the SAME grid in one call -- what the reference's products do with three calls over the same three
   arrays: most_unstable_cape_cin + mixed_layer_cape_cin (BASELINE config 5; pf.py:1557, 1651), and the most-unstable,
   100 hPa and 50 hPa mixed-layer parcels of conv_properties (pf.py:1984-2006).  parcels[i] / scalars[i] / profiles[i]
   (profiles may be NULL) describe parcel i, nparcel = 1...3.  Results are bit-identical to nparcel separate xp_cape_cin
   calls; by default that is also how the work is done (one pass per parcel, back to back on the stream).  With
   XP_OPT_FUSE_PARCELS in opts->flags, and XP_MOIST_FAMILY, dewpoint input, two surface / most-unstable / mixed-layer
   parcels and no profiles, the parcels are lifted in ONE pass (csrc/xp_multi.hpp): every level above the LCLs is read
   once and its ln p and environment virtual temperature evaluated once for both. */
int xp_cape_cin_multi(const xp_view *pressure, const xp_view *temperature, const xp_view *dewpoint,
                      int32_t nparcel, const xp_parcel *parcels, const xp_opts *opts,
                      xp_scalars_out *scalars, xp_profile_out *profiles, void *stream);

/* --- component entry points (used by the KATs and by the host-side mirrors) ------------------- */

/* pf.py:609-682 lcl: n parcels -> LCL pressure / temperature / virtual temperature. */
int xp_lcl(int64_t n, int32_t dtype, int32_t mem, const void *parcel_pressure, const void *parcel_temperature,
           const void *parcel_dewpoint, void *lcl_pressure, void *lcl_temperature,
           void *lcl_virtual_temperature, int32_t *status, void *stream);

/* pf.py:291-316 dry_lapse and pf.py:525-607 moist_lapse: parcel (ncol values) lifted to every level of
   `pressure`; out has the layout of `pressure`.  parcel_pressure NULL = level 0 (pf.py:549-550). */
int xp_dry_lapse(const xp_view *pressure, const void *parcel_temperature, const void *parcel_pressure,
                 void *out, void *stream);
int xp_moist_lapse(const xp_view *pressure, const void *parcel_temperature, const void *parcel_pressure,
                   int32_t moist_mode, void *out, void *stream);

/* pf.py:712-780 parcel_profile (no LCL level): parcel temperature and virtual temperature on the
   levels of `pressure` + LCL scalars (nullable). */
int xp_parcel_profile(const xp_view *pressure, const void *parcel_pressure, const void *parcel_temperature,
                      const void *parcel_dewpoint, int32_t moist_mode, void *temperature_out,
                      void *virtual_temperature_out, void *lcl_pressure, void *lcl_temperature,
                      void *lcl_virtual_temperature, void *stream);

/* pf.py:1066-1198 lfc_el on caller-supplied profiles (parcel / environment temperature of any kind). */
int xp_lfc_el(const xp_view *pressure, const xp_view *parcel_temperature, const xp_view *temperature,
              const void *lcl_pressure, const void *lcl_temperature, xp_scalars_out *out, void *stream);

/* pf.py:1291-1392 cape_cin_base on caller-supplied profiles and LFC / EL pressures. */
int xp_cape_cin_base(const xp_view *pressure, const xp_view *temperature, const xp_view *parcel_temperature,
                     const void *lfc_pressure, const void *el_pressure, const xp_opts *opts,
                     void *cape, void *cin, void *stream);

/* pf.py:102-135 most_unstable_parcel, pf.py:229-289 mixed_parcel: parcels only (scalars->parcel_*,
   parcel_index). */
int xp_select_parcel(const xp_view *pressure, const xp_view *temperature, const xp_view *dewpoint,
                     const xp_parcel *parcel, xp_scalars_out *out, void *stream);

/* pf.py:137-162 mixed_layer: pressure-weighted layer mean of one variable over the lowest `depth` hPa. */
int xp_mixed_layer(const xp_view *pressure, const xp_view *variable, double depth, void *out, void *stream);

/* --- SURVEY 8(f) "next" items built on the same device code ------------------------------------------ */

/* pf.py:389-445 wet_bulb_temperature (Normand's rule): for every element, lift to the LCL (pf.py:609) and come back down
   the moist adiabat to the element's own pressure (pf.py:525).  out has the layout of `pressure`. */
int xp_wet_bulb_temperature(const xp_view *pressure, const xp_view *temperature, const xp_view *dewpoint,
                            int32_t moist_mode, void *out, void *stream);

/* pf.py:1758-1811 linear_interp / pf.py:1813-1828 log_interp: value of `variable` at coordinate `at` (one value per
   column, or a single value for all when at_is_scalar) between the bracketing levels of `coords`; duplicates of a
   bracketing coordinate are averaged, no extrapolation (NaN).  log_coords != 0 interpolates in ln(coords), ln(at).
   Used by lifted_index (pf.py:1722), deep_convective_index (pf.py:1830), isobar_temperature (pf.py:2193). */
int xp_interp_level(const xp_view *coords, const xp_view *variable, const void *at, int32_t at_is_scalar,
                    int32_t log_coords, void *out, void *stream);
/* The same rule for nvar (1..4) variables at ntarget (1..4) coordinates in one pass over the column: what conv_properties
   (pf.py:1951) needs of deep_convective_index (pf.py:1830: T, Td at 850 hPa), lapse_rate (pf.py:2102: T, z at 700 and 500)
   and isobar_temperature (pf.py:2193: T at 500) is one call instead of seven.  out[v * ntarget + j] (ncol values each,
   NULL = not wanted) = variable v at coordinate at[j]. */
int xp_interp_levels(const xp_view *coords, int32_t nvar, const xp_view *const *variables, int32_t ntarget, const double *at,
                     int32_t log_coords, void *const *out, void *stream);

/* metpy.calc.dewpoint_from_specific_humidity in its MetPy 1.4.1 form, the front step of the reference's harness and
   products (parcel_test.py:262-266, pf.py:1889-1894, 1969-1974): w = q/(1-q), RH = w / w_s(p, T),
   Td = dewpoint(RH e_s(T)) [K].  Element-wise; out has the layout of `pressure`.  No reference fixture pins this
   function: parity unpinned beyond the oracle's restatement. */
int xp_dewpoint_from_specific_humidity(const xp_view *pressure, const xp_view *temperature,
                                       const xp_view *specific_humidity, void *out, void *stream);

/* pf.py:2137-2158 freezing_level_height (and melting_level_height, pf.py:2160-2191, on the wet-bulb field): the
   smallest x over all intersections (find_intersections pf.py:992-1064, linear in x) of the profile a(x) with the
   constant `value`; NaN where the profile never crosses it.  One value per column. */
int xp_crossing_level(const xp_view *x, const xp_view *a, double value, void *out, void *stream);

/* pf.py:684-710 mixing_ratio: w = RH(T, Td) * w_s(p, T) [kg/kg], element-wise (MetPy 1.4.1 forms, pf.py:698-704); out has
   the layout of `temperature`.  (virtual_temperature, pf.py:782-804, is T (1 + 0.608 w): plain arithmetic in the mirror.) */
int xp_mixing_ratio(const xp_view *temperature, const xp_view *dewpoint, const xp_view *pressure, void *out, void *stream);

/* pf.py:1951-2100 conv_properties, the reference's product bundle, in ONE call: the q -> dewpoint front step and the NaN mask,
   most-unstable (250 hPa), 100 hPa and 50 hPa mixed-layer CAPE / CIN with their lifted indices (pf.py:1722) out of the same
   passes, the three deep convective indices (pf.py:1830), the 700-500 hPa lapse rate (pf.py:2102), the 500 hPa temperature
   (pf.py:2193), freezing and melting level (pf.py:2137, 2160 with the 1/3-rule wet bulb), the 0-6 km shear (pf.py:2216) and the
   mixing ratio of the most-unstable parcel (pf.py:2053-2059); points with a NaN anywhere in pressure / temperature / humidity /
   dewpoint are blanked unless ignore_nans (pf.py:2097-2098).  Everything that is not parcel lifting is one pass over the four
   grids plus one per-point kernel (csrc/xp_bundle.hpp); scratch (the dewpoint array, per-point temporaries) is stream-ordered
   and internal.  The four (nlev, ncol) views share dtype and mem; the wind views are (nwind, ncol) on their own vertical;
   surface winds and every output are ncol values of that dtype in that mem (positive_shear: int32 0 / 1); NULL outputs are
   skipped.  opts: moist_mode and the CAPE / CIN options of xp_cape_cin (humidity is ignored: the input IS specific humidity). */
typedef struct {
    const xp_view *pressure, *temperature, *specific_humidity, *height_asl;
    const xp_view *wind_u, *wind_v, *wind_height_above_surface;
    const void *surface_wind_u, *surface_wind_v;
} xp_conv_in;
typedef struct {
    void *mu_cape, *mu_cin, *mu_mixing_ratio, *mu_lifted_index, *mu_dci;
    void *mixed_100_cape, *mixed_100_cin, *mixed_100_lifted_index, *mixed_100_dci;
    void *mixed_50_cape, *mixed_50_cin, *mixed_50_lifted_index, *mixed_50_dci;
    void *lapse_rate_700_500, *temp_500, *freezing_level, *melting_level;
    void *shear_u, *shear_v, *shear_magnitude;
    int32_t *positive_shear;
} xp_conv_out;
int xp_conv_properties(const xp_conv_in *in, const xp_opts *opts, int32_t ignore_nans, xp_conv_out *out, void *stream);

/* ---- Per-point products on top of the bundle (no parcel lifting: array arithmetic in the reference) --------------------
   n values each, of `dtype`, in `mem`; flags are int32 0 / 1. */

/* pf.py:2216-2259 wind_shear: the wind at shear_height [m] (linear interpolation in height over the (nwind, ncol) views,
   pf.py:1758 rule) minus the surface wind; outputs nullable. */
int xp_wind_shear(const xp_view *wind_u, const xp_view *wind_v, const xp_view *height, const void *surface_wind_u,
                  const void *surface_wind_v, double shear_height, void *shear_u, void *shear_v, void *shear_magnitude,
                  int32_t *positive_shear, void *stream);

/* pf.py:2261-2306 significant_hail_parameter (SPC SHIP) with the reference's validity windows. */
int xp_significant_hail_parameter(int64_t n, int32_t dtype, int32_t mem, const void *mucape, const void *mixing_ratio,
                                  const void *lapse, const void *temp_500, const void *shear, const void *flh, void *out,
                                  void *stream);

/* pf.py:2323-2407 storm_proxies from the outputs of xp_conv_properties: nine flags (each nullable) and SHIP. */
typedef struct {
    const void *mu_cape, *mu_mixing_ratio, *mixed_100_cape, *mixed_100_cin, *mixed_100_lifted_index, *mixed_100_dci;
    const void *mixed_50_cape, *mixed_50_cin, *lapse_rate_700_500, *temp_500, *freezing_level, *shear_magnitude;
    const int32_t *positive_shear;
} xp_proxies_in;
typedef struct {
    int32_t *craven2004, *kunz2007, *trapp2007, *marsh2009, *allen2011, *allen2014, *eccel2012, *mohr2013, *ship_0_1;
    void *ship;
} xp_proxies_out;
int xp_storm_proxies(int64_t n, int32_t dtype, int32_t mem, const xp_proxies_in *in, xp_proxies_out *out, void *stream);

/* ---- Array primitives of the reference's implementation -------------------------------------------------------------
   The CAPE / CIN kernels stream a column once and never build the arrays these functions return, but the reference
   exposes them (and its tests call two of them), so a caller of the reference finds them here too.  One variable per
   call; every view of a call shares shape, dtype and mem; outputs are DENSE C-order (rows, ncol) arrays of that dtype
   in that mem, allocated by the caller.  Arithmetic follows the reference's expressions operation by operation. */

/* pf.py:933-990 insert_level, one variable: out (nlev + 1, ncol) = `variable` with level_value inserted after every
   level whose coordinate is >= level_coord (an equal coordinate stays BELOW the new level, pf.py:950-954).  For the
   coordinate itself pass variable = coords, level_value = level_coord.  Rows whose coordinate is NaN come out NaN in
   every variable, and so does a value equal to fill_value (the reference's -999 trick, pf.py:962-966, 988; its assert
   that the data holds no fill_value is not evaluated). */
int xp_insert_level(const xp_view *coords, const xp_view *variable, const void *level_coord, const void *level_value,
                    double fill_value, void *out, void *stream);

/* pf.py:992-1064 find_intersections of a(x) and b(x) (b NULL = zero): six (nlev - 1, ncol) arrays, each nullable --
   all_intersect_x, all_intersect_y, increasing_x, increasing_y, decreasing_x, decreasing_y; row i describes the interval
   between levels i and i + 1 (the reference's label i + 1 on 'offset_dim'); log_x: interpolate in ln x (pf.py:1014, 1053). */
int xp_find_intersections(const xp_view *x, const xp_view *a, const xp_view *b, int32_t log_x, void *const out[6], void *stream);

/* pf.py:164-206 trapz of one variable: out[c] = sum over intervals of |dx| * mean(dat), NaN areas skipped; mask
   (nullable): (nlev - 1, ncol) bytes in the views' mem, interval i counts when non-zero. */
int xp_trapz(const xp_view *dat, const xp_view *x, const uint8_t *mask, int32_t only_positive, int32_t only_negative,
             void *out, void *stream);

/* pf.py:1200-1289 trap_around_zeros (start = 0): areas = area, dx, x, x_from, x_to, each (2 nlev - 1, ncol), nullable:
   rows 0 .. nlev-1 the areas just BEFORE a zero of y (level k and the zero in (k, k+1)), rows nlev .. 2 nlev-2 the areas
   just AFTER (the zero of interval i and level i + 1) -- the reference's concat of the two families (pf.py:1273).
   mask (nullable): (nlev, ncol) bytes, 1 where the "before" area is NaN (pf.py:1282-1287). */
int xp_trap_around_zeros(const xp_view *x, const xp_view *y, int32_t log_x, void *const areas[5], uint8_t *mask, void *stream);

/* pf.py:208-227 bound_pressure: the pressure of the column closest to bound[c] (the larger of two equally close). */
int xp_bound_pressure(const xp_view *pressure, const void *bound, void *out, void *stream);

/* pf.py:63-100 get_layer, one variable: the lowest `depth` hPa of the column (from its highest pressure), NaN outside.
   interpolate != 0: the layer top is inserted as a level (variable interpolated in ln p; the pressure variable --
   variable_is_pressure -- takes the top pressure itself): out (nlev + 1, ncol).  interpolate == 0: the top is the
   nearest existing level (bound_pressure): out (nlev, ncol). */
int xp_get_layer(const xp_view *pressure, const xp_view *variable, double depth, int32_t interpolate,
                 int32_t variable_is_pressure, void *out, void *stream);

/* pf.py:1699-1720 shift_out_nans, one variable: every column moved down by the number of leading NaNs of `name`. */
int xp_shift_out_nans(const xp_view *name, const xp_view *variable, void *out, void *stream);

/* pf.py:1517-1555 from_most_unstable_parcel (XP_PARCEL_MOST_UNSTABLE) / pf.py:1604-1649 mix_layer
   (XP_PARCEL_MIXED_LAYER): the profile re-based on its parcel.  Levels below the most-unstable parcel / inside the
   mixed layer are masked, levels left without a value in every column of the grid are dropped (dropna(how='all')),
   every column is shifted onto its first remaining level and, for the mixed layer, the parcel is put underneath.
   out_*: (nlev [+ 1 for the mixed layer], ncol), rows >= *nlev_out NaN; parcel: parcel_pressure / temperature /
   dewpoint / parcel_index of xp_scalars_out (nullable, dtype / mem of the views); level_kept (nullable, HOST, nlev
   int32): 1 for every input level that survived the drop.  *nlev_out (HOST) = rows in use.  Synchronises the stream. */
int xp_rebase_profile(const xp_view *pressure, const xp_view *temperature, const xp_view *dewpoint, const xp_parcel *parcel,
                      void *out_pressure, void *out_temperature, void *out_dewpoint, xp_scalars_out *parcel_out,
                      int32_t *level_kept, int64_t *nlev_out, void *stream);

/* pf.py:23-37 interp1d_numba = numpy.interp along the levels: at (m, ncol); xp, fp (n, ncol) with xp increasing along
   the levels (col_stride 0 shares one set of points between the columns); out (m, ncol). */
int xp_interp1d(const xp_view *at, const xp_view *xp, const xp_view *fp, void *out, void *stream);

const char *xp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* XPARCEL_H */
