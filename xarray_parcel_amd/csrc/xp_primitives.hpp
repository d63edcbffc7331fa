// xp_primitives.hpp -- the array primitives of the reference's implementation (traupach/xarray_parcel
// modules/parcel_functions.py, "pf.py") as column kernels: insert_level, find_intersections, trapz, trap_around_zeros,
// bound_pressure, get_layer, shift_out_nans, the re-basing of a profile on its most-unstable / mixed-layer parcel, and
// the 1-D interpolation of the table lookup.
//
// The CAPE / CIN kernels never build these arrays (xp_device.hpp streams a column once and carries what the
// reference's where / shift / concat expressions would produce in registers); the reference exposes the functions to
// its callers, so the library does too.  One thread = one column, lanes own adjacent columns, every level read and
// written is one coalesced request per array.  Arithmetic follows the reference's expressions operation by operation
// (no FMA contraction), so that results equal NumPy's up to the last bit of the library exp / log.
#pragma once
#include "xp_kernels.hpp"

namespace xp {

XP_DEV double sign_(double v) { return isnan_(v) ? v : (double)((v > 0.0) - (v < 0.0)); }   // np.sign
XP_DEV void st_row(const OutView &o, int f64, int64_t k, int64_t c, double v) { st(o.data, f64, k * o.ls + c * o.cs, v); }

// ---- insert_level (pf.py:933-990) for ONE variable of the dataset -----------------------------------------------------
// `cv` is the coordinate the dataset is sorted by (decreasing along the levels), `vv` the variable (the coordinate
// itself included: vv = cv, lev_v = lev_c).  The new level goes after every level whose coordinate is >= the new
// coordinate, so an existing equal coordinate stays below it (pf.py:950-954).  Rows with a NaN coordinate carry the
// fill value through the merge (pf.py:962-966: they count as "above" for any positive new coordinate) and come out
// NaN in every variable; a value equal to the fill value comes out NaN too (pf.py:988).  A NaN new coordinate matches
// neither side: every row of the output then holds the new level (pf.py:985).
template <typename T> XP_DEV void insert_level_row(int64_t j, int64_t nlev, double cc, double vc, double c0, double v0, double L,
                                                   double Lv, double fill, double &cm, double &r) {
    // (cc, vc): level j with the fill value where the coordinate is NaN (unused for j == nlev); (c0, v0): level j - 1
    double vm;
    if (j < nlev && cc >= L) { cm = cc; vm = vc; }              // pf.py:968 "below"
    else if (j >= 1 && c0 < L) { cm = c0; vm = v0; }             // pf.py:969-977 "above", shifted up one index
    else { cm = qnan(); vm = qnan(); }
    r = isnan_(cm) ? Lv : vm;                                    // pf.py:985
    if (r == fill) r = qnan();                                   // pf.py:988
}
template <typename T> __global__ __launch_bounds__(256)
void k_insert_level(View cv, View vv, int64_t nlev, int64_t ncol, const void *lev_c, const void *lev_v, double fill, OutView out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    const double L = ld1<T>(lev_c, c), Lv = ld1<T>(lev_v, c);
    double c0 = qnan(), v0 = qnan();
    for (int64_t j = 0; j <= nlev; ++j) {
        double cc = qnan(), vc = qnan();
        if (j < nlev) {
            cc = ld<T>(cv, j, c); vc = ld<T>(vv, j, c);
            if (isnan_(cc)) { cc = fill; vc = fill; }            // pf.py:966
        }
        double cm, r;
        insert_level_row<T>(j, nlev, cc, vc, c0, v0, L, Lv, fill, cm, r);
        st_row(out, f64, j, c, r);
        c0 = cc; v0 = vc;
    }
}

// ---- find_intersections (pf.py:992-1064) ------------------------------------------------------------------------------
// Row i of the six outputs (all / increasing / decreasing x and y; dense (nlev - 1, ncol), each nullable) describes the
// interval between levels i and i + 1 (the reference's label i + 1 on 'offset_dim').  An interval is examined when
// sign(a - b) changes or is NaN at either end (pf.py:1019-1022); `bv.data == nullptr` stands for b = 0.
struct Intersection { double x, y, sign_change; };
XP_DEV Intersection intersect(double x0, double x1, double a0, double a1, double b0, double b1) {
#pragma clang fp contract(off)
    Intersection r; r.x = r.y = r.sign_change = qnan();
    const double dy0 = a0 - b0, dy1 = a1 - b1;
    const double diffs = sign_(dy1) - sign_(dy0);               // pf.py:1019
    if (diffs == 0.0) return r;                                  // pf.py:1022 (NaN counts as a change)
    r.sign_change = sign_(dy1);                                  // pf.py:1031
    r.x = (dy1 * x0 - dy0 * x1) / (dy1 - dy0);                   // pf.py:1046
    r.y = ((r.x - x0) / (x1 - x0)) * (a1 - a0) + a0;             // pf.py:1050
    return r;
}
struct SixOut { void *p[6]; };
template <typename T> __global__ __launch_bounds__(256)
void k_find_intersections(View xv, View av, View bv, int64_t nlev, int64_t ncol, int log_x, SixOut o) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    double x0 = qnan(), a0 = qnan(), b0 = qnan();
    for (int64_t k = 0; k < nlev; ++k) {
        double x1 = ld<T>(xv, k, c), a1 = ld<T>(av, k, c), b1 = bv.data ? ld<T>(bv, k, c) : 0.0;
        if (log_x) x1 = log(x1);
        if (k > 0) {
            Intersection r = intersect(x0, x1, a0, a1, b0, b1);
            if (log_x) r.x = exp(r.x);                           // pf.py:1053
            const int64_t i = (k - 1) * ncol + c;
            const bool inc = r.sign_change > 0.0, dec = r.sign_change < 0.0;
            st(o.p[0], f64, i, r.x); st(o.p[1], f64, i, r.y);
            st(o.p[2], f64, i, inc ? r.x : qnan()); st(o.p[3], f64, i, inc ? r.y : qnan());
            st(o.p[4], f64, i, dec ? r.x : qnan()); st(o.p[5], f64, i, dec ? r.y : qnan());
        }
        x0 = x1; a0 = a1; b0 = b1;
    }
}

// ---- trapz (pf.py:164-206) of one variable -----------------------------------------------------------------------------
// sum over the intervals of |dx| * mean(y), skipping NaN areas; `mask` (dense (nlev - 1, ncol) bytes, nullable) keeps
// interval i when non-zero; only_positive / only_negative keep areas of that sign.
template <typename T> __global__ __launch_bounds__(256)
void k_trapz(View dv, View xv, const uint8_t *mask, int64_t nlev, int64_t ncol, int only_pos, int only_neg, void *out) {
#pragma clang fp contract(off)
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double s = 0.0, d0 = qnan(), x0 = qnan();
    for (int64_t k = 0; k < nlev; ++k) {
        double d1 = ld<T>(dv, k, c), x1 = ld<T>(xv, k, c);
        if (k > 0 && (!mask || mask[(k - 1) * ncol + c])) {
            double area = fabs(x1 - x0) * ((d0 + d1) * 0.5);     // pf.py:186-198
            if (only_pos && !(area > 0.0)) area = qnan();
            if (only_neg && !(area < 0.0)) area = qnan();
            if (!isnan_(area)) s += area;
        }
        d0 = d1; x0 = x1;
    }
    st(out, sizeof(T) == 8, c, s);
}

// ---- trap_around_zeros (pf.py:1200-1289, start = 0) --------------------------------------------------------------------
// areas: five dense (2 nlev - 1, ncol) arrays -- area, dx, x, x_from, x_to; rows 0 .. nlev-1 are the reference's
// "before zeros" family (level k just before a zero of y in (k, k+1); row nlev-1 is always NaN), rows nlev .. 2 nlev-2
// the "after zeros" family (level i+1 just after the zero of interval i).  mask: (nlev, ncol) bytes, 1 where the
// "before" area is NaN (the intervals an integration along x still has to count, pf.py:1282-1287).
struct FiveOut { void *p[5]; };
template <typename T> __global__ __launch_bounds__(256)
void k_trap_around_zeros(View xv, View yv, int64_t nlev, int64_t ncol, int log_x, FiveOut o, uint8_t *mask) {
#pragma clang fp contract(off)
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    auto put = [&](int64_t row, double y, double X, double zx, bool valid) {
        double area = qnan(), xm = qnan(), adx = qnan();
        if (valid) {
            const double dx = X - zx;                            // pf.py:1254
            adx = fabs(dx);
            area = (y / 2.0) * adx;                              // pf.py:1251, 1257
            xm = X - dx / 2.0;                                   // pf.py:1258
        }
        const int64_t i = row * ncol + c;
        st(o.p[0], f64, i, area); st(o.p[1], f64, i, adx); st(o.p[2], f64, i, xm);
        st(o.p[3], f64, i, xm - adx / 2.0); st(o.p[4], f64, i, xm + adx / 2.0);   // pf.py:1276-1277
        return area;
    };
    double X0 = qnan(), y0 = qnan();
    for (int64_t k = 0; k < nlev; ++k) {
        const double x1 = ld<T>(xv, k, c), y1 = ld<T>(yv, k, c);
        const double X1 = log_x ? log(x1) : x1;
        if (k > 0) {
            Intersection r = intersect(X0, X1, y0, y1, 0.0, 0.0);
            double zx = r.x;
            if (log_x) zx = log(exp(zx));                        // pf.py:1053, 1236
            const bool valid = !isnan_(r.y);                     // pf.py:1240
            const double ab = put(k - 1, y0, X0, zx, valid);
            put(nlev + k - 1, y1, X1, zx, valid);
            if (mask) mask[(k - 1) * ncol + c] = isnan_(ab) ? 1 : 0;
        }
        X0 = X1; y0 = y1;
    }
    put(nlev - 1, qnan(), qnan(), qnan(), false);
    if (mask) mask[(nlev - 1) * ncol + c] = 1;
}

// ---- bound_pressure (pf.py:208-227) ------------------------------------------------------------------------------------
template <typename T> XP_DEV double bound_pressure_col(const View &pv, int64_t nlev, int64_t c, double bound) {
    double dmin = qnan(), best = qnan();
    for (int64_t k = 0; k < nlev; ++k) {
        const double p = ld<T>(pv, k, c), d = fabs(p - bound);
        if (isnan_(d)) continue;
        if (!(d >= dmin)) { dmin = d; best = p; }                // strictly closer (or the first)
        else if (d == dmin && p > best) best = p;                // equally distant: the larger pressure
    }
    return best;
}
template <typename T> __global__ __launch_bounds__(256)
void k_bound_pressure(View pv, int64_t nlev, int64_t ncol, const void *bound, void *out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    st(out, sizeof(T) == 8, c, bound_pressure_col<T>(pv, nlev, c, ld1<T>(bound, c)));
}

// linear_interp (pf.py:1758-1811) of one variable at one coordinate, optionally in ln(coords) (log_interp pf.py:1813):
// k_interp_level's rule as a device function
template <typename T> XP_DEV double interp_column(const View &cv, const View &xv, int64_t nlev, int64_t c, double at, bool log_coords) {
#pragma clang fp contract(off)
    if (log_coords) at = clog(at);
    double cb = qnan(), ca = qnan(), sb = 0.0, sa = 0.0;
    int nb = 0, na = 0;
    for (int64_t k = 0; k < nlev; ++k) {
        double cc = ld<T>(cv, k, c), x = ld<T>(xv, k, c);
        if (log_coords) cc = clog(cc);
        if (isnan_(cc)) continue;
        if (cc >= at) {
            if (!(cc >= cb)) { cb = cc; sb = 0.0; nb = 0; }
            if (cc == cb && !isnan_(x)) { sb += x; ++nb; }
        }
        if (cc <= at) {
            if (!(cc <= ca)) { ca = cc; sa = 0.0; na = 0; }
            if (cc == ca && !isnan_(x)) { sa += x; ++na; }
        }
    }
    const double xb = nb ? sb / (double)nb : qnan(), xa = na ? sa / (double)na : qnan();
    const double res = xb + (xa - xb) * ((at - cb) / (ca - cb));
    return (xb == xa) ? xb : res;
}

// ---- get_layer (pf.py:63-100) for ONE variable -------------------------------------------------------------------------
// The layer from the highest pressure of the column up to `depth` hPa above it; everything outside is NaN.
// interpolate != 0: the layer top (bottom - depth) is inserted as a level of its own, the variable there interpolated
// in ln p (the pressure variable takes the top pressure itself, pf.py:86): nlev + 1 output rows.  interpolate == 0: the
// top is the existing level closest to bottom - depth (bound_pressure): nlev output rows.
template <typename T> __global__ __launch_bounds__(256)
void k_get_layer(View pv, View vv, int64_t nlev, int64_t ncol, double depth, int interpolate, int is_pressure, OutView out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    double bottom = qnan();
    for (int64_t k = 0; k < nlev; ++k) { const double p = ld<T>(pv, k, c); if (!isnan_(p) && !(p <= bottom)) bottom = p; }   // pf.py:80
    if (!interpolate) {
        const double top = bound_pressure_col<T>(pv, nlev, c, bottom - depth);                  // pf.py:92-94
        for (int64_t k = 0; k < nlev; ++k) {
            const double p = ld<T>(pv, k, c);
            st_row(out, f64, k, c, (p <= bottom && p >= top) ? ld<T>(vv, k, c) : qnan());         // pf.py:97-98
        }
        return;
    }
    const double top = bottom - depth, fill = -999.0;
    const double Lv = is_pressure ? top : interp_column<T>(pv, vv, nlev, c, top, true);         // pf.py:84-86
    double c0 = qnan(), v0 = qnan();
    for (int64_t j = 0; j <= nlev; ++j) {
        double cc = qnan(), vc = qnan();
        if (j < nlev) {
            cc = ld<T>(pv, j, c); vc = ld<T>(vv, j, c);
            if (isnan_(cc)) { cc = fill; vc = fill; }
        }
        double cm, r, pm, rp;
        insert_level_row<T>(j, nlev, cc, vc, c0, v0, top, Lv, fill, cm, r);
        insert_level_row<T>(j, nlev, cc, cc, c0, c0, top, top, fill, pm, rp);                     // the merged pressure of this row
        st_row(out, f64, j, c, (rp <= bottom && rp >= top) ? r : qnan());
        c0 = cc; v0 = vc;
    }
}

// ---- shift_out_nans (pf.py:1699-1720) for ONE variable -----------------------------------------------------------------
// every column moves down by the number of leading NaNs of `name` in that column; what is shifted in at the top is NaN
template <typename T> __global__ __launch_bounds__(256)
void k_shift_out_nans(View nv, View vv, int64_t nlev, int64_t ncol, OutView out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    int64_t s = 0;
    while (s < nlev && isnan_(ld<T>(nv, s, c))) ++s;
    for (int64_t j = 0; j < nlev; ++j) st_row(out, sizeof(T) == 8, j, c, j + s < nlev ? ld<T>(vv, j + s, c) : qnan());
}

// ---- from_most_unstable_parcel (pf.py:1517-1555) / mix_layer (pf.py:1604-1649) -----------------------------------------
// The profile re-based on its parcel: levels below the most-unstable parcel / inside the mixed layer are masked
// (where), levels left without a value in ANY column of the grid are dropped (dropna(how='all')), every column is
// shifted down onto its first remaining level (shift_out_nans) and, for the mixed layer, the parcel is put underneath.
// Pass 1: the parcel, the column's pressure threshold, and which levels survive somewhere in the grid.
template <typename T, int PMODE> __global__ __launch_bounds__(256)
void k_rebase_select(CapeArgs a, double *thr, int32_t *level_any) {
    __shared__ double s_es[LDS_TAB];
    const double *es = stage_es_table(a.es_tab, s_es);
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncol) return;
    Parcel pc = (PMODE == PM_MU) ? select_mu<T, false>(a, c, es, a.depth) : select_ml<T, false>(a, c, es, a.depth);
    st(a.s.par_p, a.s.f64, c, pc.p); st(a.s.par_t, a.s.f64, c, pc.t); st(a.s.par_td, a.s.f64, c, pc.td);
    sti(a.s.parcel_idx, c, pc.idx);
    double t = pc.p;                                                       // MU: keep p <= p_parcel (pf.py:1551)
    if (PMODE == PM_ML) {                                                  // ML: keep p < max(p) - depth (pf.py:1636)
        double pmax = qnan();
        for (int64_t k = 0; k < a.nlev; ++k) { const double p = ld<T>(a.p, k, c); if (!isnan_(p) && !(p <= pmax)) pmax = p; }
        t = pmax - a.depth;
    }
    thr[c] = t;
    for (int64_t k = 0; k < a.nlev; ++k) {
        const double p = ld<T>(a.p, k, c);
        if (PMODE == PM_MU ? (p <= t) : (p < t)) level_any[k] = 1;         // (same value from every thread)
    }
}
// Pass 2: compaction.  Output rows: [parcel row for ML] + the surviving levels, NaN padding above.
struct RebaseOut { void *p, *t, *td; };
template <typename T, int PMODE> __global__ __launch_bounds__(256)
void k_rebase_write(CapeArgs a, const double *thr, const int32_t *level_any, int64_t nrow_out, RebaseOut o) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncol) return;
    const int f64 = sizeof(T) == 8;
    const double t = thr[c];
    int64_t j = 0;
    if (PMODE == PM_ML) {                                                  // pf.py:1641-1644
        st(o.p, f64, c, ld1<T>(a.s.par_p, c)); st(o.t, f64, c, ld1<T>(a.s.par_t, c)); st(o.td, f64, c, ld1<T>(a.s.par_td, c));
        j = 1;
    }
    bool leading = true;
    for (int64_t k = 0; k < a.nlev; ++k) {
        if (!level_any[k]) continue;                                       // dropna(dim, how='all')
        const double p = ld<T>(a.p, k, c);
        const bool keep = PMODE == PM_MU ? (p <= t) : (p < t);
        if (leading && !keep) continue;                                    // shift_out_nans: leading NaN pressures
        leading = false;
        const int64_t i = j * a.ncol + c;
        st(o.p, f64, i, keep ? p : qnan()); st(o.t, f64, i, keep ? ld<T>(a.t, k, c) : qnan());
        st(o.td, f64, i, keep ? ld<T>(a.td, k, c) : qnan());
        ++j;
    }
    for (; j < nrow_out; ++j) {
        const int64_t i = j * a.ncol + c;
        st(o.p, f64, i, qnan()); st(o.t, f64, i, qnan()); st(o.td, f64, i, qnan());
    }
}

// ---- interp1d_numba (pf.py:23-37): numpy.interp along the levels -------------------------------------------------------
// at: (m, ncol); xp / fp: (n, ncol) known points, xp increasing along the levels (col_stride 0 shares one set of points
// between all columns, as the table lookup does with its pressure axis); out: dense (m, ncol).  numpy.interp's rules:
// below xp[0] -> fp[0], above xp[n-1] -> fp[n-1], NaN -> NaN, on a knot -> its value.
template <typename T> __global__ __launch_bounds__(256)
void k_interp1d(View atv, View xpv, View fpv, int64_t m, int64_t n, int64_t ncol, void *out) {
#pragma clang fp contract(off)
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    const double x_lo = ld<T>(xpv, 0, c), x_hi = ld<T>(xpv, n - 1, c);
    for (int64_t i = 0; i < m; ++i) {
        const double x = ld<T>(atv, i, c);
        double r;
        if (isnan_(x)) r = x;
        else if (x <= x_lo) r = ld<T>(fpv, 0, c);
        else if (x >= x_hi) r = ld<T>(fpv, n - 1, c);
        else {
            int64_t lo = 0, hi = n - 1;                                    // xp[lo] <= x < xp[hi]
            while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (ld<T>(xpv, mid, c) <= x) lo = mid; else hi = mid; }
            const double xl = ld<T>(xpv, lo, c), fl = ld<T>(fpv, lo, c), fh = ld<T>(fpv, hi, c), xh = ld<T>(xpv, hi, c);
            const double slope = (fh - fl) / (xh - xl);
            r = slope * (x - xl) + fl;
            if (x == xl) r = fl;                                           // on a knot: its value, whatever the slope
            else if (isnan_(r)) { r = slope * (x - xh) + fh; if (isnan_(r) && fl == fh) r = fl; }   // numpy's fall-backs for infinite values
        }
        st(out, f64, i * ncol + c, r);
    }
}

}  // namespace xp
