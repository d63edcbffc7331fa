// xp_primitives_abi.hpp -- C ABI of the array primitives (kernels: xp_primitives.hpp; declarations and the reference
// functions they replace: include/xparcel.h).  Included by xparcel.hip, whose staging helpers it uses.
#include "xp_primitives.hpp"

namespace {

// launch a (T)-templated column kernel for the dtype of view `v` over `n` threads
#define XP_LAUNCH_T(v, n, st, kernel, ...)                                                                          \
    do {                                                                                                            \
        if ((n) > 0) {                                                                                              \
            if ((v)->dtype == XP_F64) hipLaunchKernelGGL((xp::kernel<double>), dim3(blocks(n)), dim3(256), 0, (st).s, __VA_ARGS__); \
            else hipLaunchKernelGGL((xp::kernel<float>), dim3(blocks(n)), dim3(256), 0, (st).s, __VA_ARGS__);       \
        }                                                                                                           \
    } while (0)

size_t rows_bytes(const xp_view *v, int64_t rows) { return (size_t)rows * (size_t)v->ncol * esize(v->dtype); }
xp::OutView dense_out(void *d, int64_t ncol) { xp::OutView o; o.data = d; o.ls = ncol; o.cs = 1; return o; }

}  // namespace

extern "C" {

int xp_insert_level(const xp_view *coords, const xp_view *variable, const void *level_coord, const void *level_value,
                    double fill_value, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(coords, "coords")) || (rc = check_view(variable, "variable")) || (rc = same_shape(coords, variable, "coords/variable"))) return rc;
    if (!level_coord || !level_value || !out) return fail(XP_E_ARG, "xp_insert_level: null argument");
    Stager st(stream);
    xp::View cv, vv;
    const void *lc, *lv;
    void *od;
    const size_t cb = rows_bytes(coords, 1);
    if ((rc = stage_view(st, coords, &cv)) || (rc = stage_view(st, variable, &vv)) || (rc = st.in(level_coord, cb, coords->mem, &lc)) ||
        (rc = st.in(level_value, cb, coords->mem, &lv)) || (rc = st.out(out, rows_bytes(coords, coords->nlev + 1), coords->mem, &od))) return rc;
    XP_LAUNCH_T(coords, coords->ncol, st, k_insert_level, cv, vv, coords->nlev, coords->ncol, lc, lv, fill_value, dense_out(od, coords->ncol));
    return st.finish();
}

int xp_find_intersections(const xp_view *x, const xp_view *a, const xp_view *b, int32_t log_x, void *const out[6], void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(x, "x")) || (rc = check_view(a, "a")) || (rc = same_shape(x, a, "x/a"))) return rc;
    if (b && ((rc = check_view(b, "b")) || (rc = same_shape(x, b, "x/b")))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_find_intersections: null output list");
    Stager st(stream);
    xp::View xv, av, bv;
    bv.data = nullptr; bv.ls = bv.cs = 0;
    xp::SixOut o;
    if ((rc = stage_view(st, x, &xv)) || (rc = stage_view(st, a, &av)) || (b && (rc = stage_view(st, b, &bv)))) return rc;
    for (int i = 0; i < 6; ++i) if ((rc = st.out(out[i], rows_bytes(x, x->nlev - 1), x->mem, &o.p[i]))) return rc;
    XP_LAUNCH_T(x, x->ncol, st, k_find_intersections, xv, av, bv, x->nlev, x->ncol, (int)log_x, o);
    return st.finish();
}

int xp_trapz(const xp_view *dat, const xp_view *x, const uint8_t *mask, int32_t only_positive, int32_t only_negative,
             void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(dat, "dat")) || (rc = check_view(x, "x")) || (rc = same_shape(dat, x, "dat/x"))) return rc;
    if (only_positive && only_negative)
        return fail(XP_E_ARG, "Only negative OR positive regions can be included in trapz.");     // pf.py:200
    if (!out) return fail(XP_E_ARG, "xp_trapz: null output");
    Stager st(stream);
    xp::View dv, xv;
    const void *dm;
    void *od;
    if ((rc = stage_view(st, dat, &dv)) || (rc = stage_view(st, x, &xv)) ||
        (rc = st.in(mask, (size_t)(dat->nlev - 1) * (size_t)dat->ncol, dat->mem, &dm)) || (rc = st.out(out, rows_bytes(dat, 1), dat->mem, &od))) return rc;
    XP_LAUNCH_T(dat, dat->ncol, st, k_trapz, dv, xv, (const uint8_t *)dm, dat->nlev, dat->ncol, (int)only_positive, (int)only_negative, od);
    return st.finish();
}

int xp_trap_around_zeros(const xp_view *x, const xp_view *y, int32_t log_x, void *const areas[5], uint8_t *mask, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(x, "x")) || (rc = check_view(y, "y")) || (rc = same_shape(x, y, "x/y"))) return rc;
    if (!areas) return fail(XP_E_ARG, "xp_trap_around_zeros: null output list");
    Stager st(stream);
    xp::View xv, yv;
    xp::FiveOut o;
    void *dm;
    if ((rc = stage_view(st, x, &xv)) || (rc = stage_view(st, y, &yv))) return rc;
    for (int i = 0; i < 5; ++i) if ((rc = st.out(areas[i], rows_bytes(x, 2 * x->nlev - 1), x->mem, &o.p[i]))) return rc;
    if ((rc = st.out(mask, (size_t)x->nlev * (size_t)x->ncol, x->mem, &dm))) return rc;
    XP_LAUNCH_T(x, x->ncol, st, k_trap_around_zeros, xv, yv, x->nlev, x->ncol, (int)log_x, o, (uint8_t *)dm);
    return st.finish();
}

int xp_bound_pressure(const xp_view *pressure, const void *bound, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(pressure, "pressure"))) return rc;
    if (!bound || !out) return fail(XP_E_ARG, "xp_bound_pressure: null argument");
    Stager st(stream);
    xp::View pv;
    const void *db;
    void *od;
    if ((rc = stage_view(st, pressure, &pv)) || (rc = st.in(bound, rows_bytes(pressure, 1), pressure->mem, &db)) ||
        (rc = st.out(out, rows_bytes(pressure, 1), pressure->mem, &od))) return rc;
    XP_LAUNCH_T(pressure, pressure->ncol, st, k_bound_pressure, pv, pressure->nlev, pressure->ncol, db, od);
    return st.finish();
}

int xp_get_layer(const xp_view *pressure, const xp_view *variable, double depth, int32_t interpolate,
                 int32_t variable_is_pressure, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(pressure, "pressure")) || (rc = check_view(variable, "variable")) ||
        (rc = same_shape(pressure, variable, "pressure/variable"))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_get_layer: null output");
    Stager st(stream);
    xp::View pv, vv;
    void *od;
    if ((rc = stage_view(st, pressure, &pv)) || (rc = stage_view(st, variable, &vv)) ||
        (rc = st.out(out, rows_bytes(pressure, pressure->nlev + (interpolate ? 1 : 0)), pressure->mem, &od))) return rc;
    XP_LAUNCH_T(pressure, pressure->ncol, st, k_get_layer, pv, vv, pressure->nlev, pressure->ncol, depth, (int)interpolate,
                (int)variable_is_pressure, dense_out(od, pressure->ncol));
    return st.finish();
}

int xp_shift_out_nans(const xp_view *name, const xp_view *variable, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(name, "name")) || (rc = check_view(variable, "variable")) || (rc = same_shape(name, variable, "name/variable"))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_shift_out_nans: null output");
    Stager st(stream);
    xp::View nv, vv;
    void *od;
    if ((rc = stage_view(st, name, &nv)) || (rc = stage_view(st, variable, &vv)) ||
        (rc = st.out(out, rows_bytes(name, name->nlev), name->mem, &od))) return rc;
    XP_LAUNCH_T(name, name->ncol, st, k_shift_out_nans, nv, vv, name->nlev, name->ncol, dense_out(od, name->ncol));
    return st.finish();
}

int xp_rebase_profile(const xp_view *p, const xp_view *t, const xp_view *td, const xp_parcel *parcel, void *out_p, void *out_t,
                      void *out_td, xp_scalars_out *parcel_out, int32_t *level_kept, int64_t *nlev_out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (!parcel || (parcel->mode != XP_PARCEL_MOST_UNSTABLE && parcel->mode != XP_PARCEL_MIXED_LAYER))
        return fail(XP_E_ARG, "xp_rebase_profile: mode must be most-unstable or mixed-layer");
    if (!nlev_out) return fail(XP_E_ARG, "xp_rebase_profile: null nlev_out");
    Stager st(stream);
    xp::CapeArgs a;
    if ((rc = fill_common(st, p, t, td, parcel, nullptr, &a))) return rc;
    if (parcel_out && (parcel_out->dtype != p->dtype || parcel_out->mem != p->mem))
        return fail(XP_E_ARG, "xp_rebase_profile: parcel_out must have the dtype and mem of the views");
    const bool ml = parcel->mode == XP_PARCEL_MIXED_LAYER;
    const int64_t nlev = p->nlev, ncol = p->ncol, rows = nlev + (ml ? 1 : 0);
    xp_scalars_out none;
    memset(&none, 0, sizeof(none));
    none.dtype = p->dtype; none.mem = p->mem;
    if ((rc = stage_scalars(st, parcel_out ? parcel_out : &none, ncol, &a.s))) return rc;
    // pass 2 reads the mixed-layer parcel back: give it scratch where the caller did not ask for it
    void **need[3] = {&a.s.par_p, &a.s.par_t, &a.s.par_td};
    for (int i = 0; i < 3; ++i)
        if (!*need[i]) { HIP_TRY(hipMallocAsync(need[i], rows_bytes(p, 1) + 8, st.s)); st.scratch.push_back(*need[i]); }
    double *thr = nullptr;
    int32_t *any = nullptr;
    HIP_TRY(hipMallocAsync((void **)&thr, (size_t)(ncol + 1) * 8, st.s)); st.scratch.push_back(thr);
    HIP_TRY(hipMallocAsync((void **)&any, (size_t)nlev * 4, st.s)); st.scratch.push_back(any);
    HIP_TRY(hipMemsetAsync(any, 0, (size_t)nlev * 4, st.s));
    xp::RebaseOut o;
    if ((rc = st.out(out_p, rows_bytes(p, rows), p->mem, &o.p)) || (rc = st.out(out_t, rows_bytes(p, rows), p->mem, &o.t)) ||
        (rc = st.out(out_td, rows_bytes(p, rows), p->mem, &o.td))) return rc;
    if (ncol) {
        dim3 gr(blocks(ncol)), bl(256);
        if (p->dtype == XP_F64) {
            if (ml) hipLaunchKernelGGL((xp::k_rebase_select<double, xp::PM_ML>), gr, bl, 0, st.s, a, thr, any);
            else hipLaunchKernelGGL((xp::k_rebase_select<double, xp::PM_MU>), gr, bl, 0, st.s, a, thr, any);
        } else {
            if (ml) hipLaunchKernelGGL((xp::k_rebase_select<float, xp::PM_ML>), gr, bl, 0, st.s, a, thr, any);
            else hipLaunchKernelGGL((xp::k_rebase_select<float, xp::PM_MU>), gr, bl, 0, st.s, a, thr, any);
        }
    }
    std::vector<int32_t> kept((size_t)nlev, 0);
    HIP_TRY(hipMemcpyAsync(kept.data(), any, (size_t)nlev * 4, hipMemcpyDeviceToHost, st.s));
    HIP_TRY(hipStreamSynchronize(st.s));                                   // the number of surviving levels shapes the output
    int64_t nkept = 0;
    for (int64_t k = 0; k < nlev; ++k) nkept += kept[k] != 0;
    if (level_kept) memcpy(level_kept, kept.data(), (size_t)nlev * 4);
    *nlev_out = nkept + (ml ? 1 : 0);
    if (ncol) {
        dim3 gr(blocks(ncol)), bl(256);
        if (p->dtype == XP_F64) {
            if (ml) hipLaunchKernelGGL((xp::k_rebase_write<double, xp::PM_ML>), gr, bl, 0, st.s, a, thr, any, rows, o);
            else hipLaunchKernelGGL((xp::k_rebase_write<double, xp::PM_MU>), gr, bl, 0, st.s, a, thr, any, rows, o);
        } else {
            if (ml) hipLaunchKernelGGL((xp::k_rebase_write<float, xp::PM_ML>), gr, bl, 0, st.s, a, thr, any, rows, o);
            else hipLaunchKernelGGL((xp::k_rebase_write<float, xp::PM_MU>), gr, bl, 0, st.s, a, thr, any, rows, o);
        }
    }
    if ((rc = st.finish())) return rc;
    HIP_TRY(hipStreamSynchronize(st.s));
    return 0;
}

int xp_interp1d(const xp_view *at, const xp_view *xp_, const xp_view *fp, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(at, "at")) || (rc = check_view(xp_, "xp")) || (rc = check_view(fp, "fp"))) return rc;
    if (xp_->dtype != at->dtype || fp->dtype != at->dtype || xp_->nlev != fp->nlev ||
        (xp_->ncol != at->ncol && xp_->ncol != 1) || (fp->ncol != at->ncol && fp->ncol != 1))
        return fail(XP_E_ARG, "xp_interp1d: xp / fp must be (n, ncol) or (n, 1) of the dtype of `at`");
    if (!out) return fail(XP_E_ARG, "xp_interp1d: null output");
    Stager st(stream);
    xp::View av, xv, fv;
    void *od;
    if ((rc = stage_view(st, at, &av)) || (rc = stage_view(st, xp_, &xv)) || (rc = stage_view(st, fp, &fv)) ||
        (rc = st.out(out, rows_bytes(at, at->nlev), at->mem, &od))) return rc;
    if (xp_->ncol == 1 && at->ncol != 1) { xv.cs = 0; xv.ls = xp_->mem == XP_MEM_HOST ? 1 : xp_->lev_stride; }   // one set of points for all columns
    if (fp->ncol == 1 && at->ncol != 1) { fv.cs = 0; fv.ls = fp->mem == XP_MEM_HOST ? 1 : fp->lev_stride; }
    XP_LAUNCH_T(at, at->ncol, st, k_interp1d, av, xv, fv, at->nlev, xp_->nlev, at->ncol, od);
    return st.finish();
}


// ---- per-point products on top of the bundle (kernels: xp_bundle.hpp) -------------------------------------------------------
int xp_wind_shear(const xp_view *wind_u, const xp_view *wind_v, const xp_view *height, const void *surface_wind_u,
                  const void *surface_wind_v, double shear_height, void *shear_u, void *shear_v, void *shear_magnitude,
                  int32_t *positive_shear, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(wind_u, "wind_u")) || (rc = check_view(wind_v, "wind_v")) || (rc = check_view(height, "height")) ||
        (rc = same_shape(wind_u, wind_v, "wind_u/wind_v")) || (rc = same_shape(wind_u, height, "wind_u/height"))) return rc;
    if (!surface_wind_u || !surface_wind_v) return fail(XP_E_ARG, "xp_wind_shear: null surface wind");
    Stager st(stream);
    xp::ShearArgs a;
    memset(&a, 0, sizeof(a));
    const size_t cb = rows_bytes(wind_u, 1);
    void *pos = nullptr;
    if ((rc = stage_view(st, wind_u, &a.u)) || (rc = stage_view(st, wind_v, &a.v)) || (rc = stage_view(st, height, &a.h)) ||
        (rc = st.in(surface_wind_u, cb, wind_u->mem, &a.sfc_u)) || (rc = st.in(surface_wind_v, cb, wind_u->mem, &a.sfc_v)) ||
        (rc = st.out(shear_u, cb, wind_u->mem, &a.shear_u)) || (rc = st.out(shear_v, cb, wind_u->mem, &a.shear_v)) ||
        (rc = st.out(shear_magnitude, cb, wind_u->mem, &a.shear_mag)) ||
        (rc = st.out(positive_shear, (size_t)wind_u->ncol * 4, wind_u->mem, &pos))) return rc;
    a.positive_shear = (int32_t *)pos;
    a.nwind = wind_u->nlev; a.ncol = wind_u->ncol; a.shear_height = shear_height;
    XP_LAUNCH_T(wind_u, wind_u->ncol, st, k_wind_shear, a);
    return st.finish();
}

int xp_significant_hail_parameter(int64_t n, int32_t dtype, int32_t mem, const void *mucape, const void *mixing_ratio,
                                  const void *lapse, const void *temp_500, const void *shear, const void *flh, void *out,
                                  void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (n < 0 || (dtype != XP_F32 && dtype != XP_F64)) return fail(XP_E_ARG, "xp_significant_hail_parameter: bad n / dtype");
    if (!mucape || !mixing_ratio || !lapse || !temp_500 || !shear || !flh || !out) return fail(XP_E_ARG, "xp_significant_hail_parameter: null argument");
    Stager st(stream);
    const size_t b = (size_t)n * esize(dtype);
    const void *in[6];
    const void *src[6] = {mucape, mixing_ratio, lapse, temp_500, shear, flh};
    void *od;
    for (int i = 0; i < 6; ++i) if ((rc = st.in(src[i], b, mem, &in[i]))) return rc;
    if ((rc = st.out(out, b, mem, &od))) return rc;
    if (n) {
        if (dtype == XP_F64) hipLaunchKernelGGL((xp::k_ship<double>), dim3(blocks(n)), dim3(256), 0, st.s, n, in[0], in[1], in[2], in[3], in[4], in[5], od);
        else hipLaunchKernelGGL((xp::k_ship<float>), dim3(blocks(n)), dim3(256), 0, st.s, n, in[0], in[1], in[2], in[3], in[4], in[5], od);
    }
    return st.finish();
}

int xp_storm_proxies(int64_t n, int32_t dtype, int32_t mem, const xp_proxies_in *in, xp_proxies_out *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (n < 0 || (dtype != XP_F32 && dtype != XP_F64)) return fail(XP_E_ARG, "xp_storm_proxies: bad n / dtype");
    if (!in || !out) return fail(XP_E_ARG, "xp_storm_proxies: null argument");
    const void *src[12] = {in->mu_cape, in->mu_mixing_ratio, in->mixed_100_cape, in->mixed_100_cin, in->mixed_100_lifted_index,
                           in->mixed_100_dci, in->mixed_50_cape, in->mixed_50_cin, in->lapse_rate_700_500, in->temp_500,
                           in->freezing_level, in->shear_magnitude};
    for (int i = 0; i < 12; ++i) if (!src[i]) return fail(XP_E_ARG, "xp_storm_proxies: null input %d", i);
    if (!in->positive_shear) return fail(XP_E_ARG, "xp_storm_proxies: null positive_shear");
    Stager st(stream);
    xp::ProxiesArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    const size_t b = (size_t)n * esize(dtype);
    const void **dst[12] = {&a.mu_cape, &a.mu_mixing_ratio, &a.mixed_100_cape, &a.mixed_100_cin, &a.mixed_100_lifted_index,
                            &a.mixed_100_dci, &a.mixed_50_cape, &a.mixed_50_cin, &a.lapse_rate_700_500, &a.temp_500,
                            &a.freezing_level, &a.shear_magnitude};
    for (int i = 0; i < 12; ++i) if ((rc = st.in(src[i], b, mem, dst[i]))) return rc;
    const void *ps;
    if ((rc = st.in(in->positive_shear, (size_t)n * 4, mem, &ps))) return rc;
    a.positive_shear = (const int32_t *)ps;
    int32_t *flags[9] = {out->craven2004, out->kunz2007, out->trapp2007, out->marsh2009, out->allen2011, out->allen2014,
                         out->eccel2012, out->mohr2013, out->ship_0_1};
    for (int i = 0; i < 9; ++i) {
        void *d;
        if ((rc = st.out(flags[i], (size_t)n * 4, mem, &d))) return rc;
        a.proxy[i] = (int32_t *)d;
    }
    if ((rc = st.out(out->ship, b, mem, &a.ship))) return rc;
    if (n) {
        if (dtype == XP_F64) hipLaunchKernelGGL((xp::k_storm_proxies<double>), dim3(blocks(n)), dim3(256), 0, st.s, a);
        else hipLaunchKernelGGL((xp::k_storm_proxies<float>), dim3(blocks(n)), dim3(256), 0, st.s, a);
    }
    return st.finish();
}

}  // extern "C"
