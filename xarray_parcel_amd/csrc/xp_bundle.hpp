// xp_bundle.hpp -- the column and per-point arithmetic of the reference's product bundle conv_properties (pf.py:1951-2100)
// that is NOT parcel lifting, in two kernels, so that the bundle issues no array arithmetic outside the library:
//   k_conv_columns  one pass over (pressure, temperature, specific humidity, height): the q -> dewpoint front step
//                   (pf.py:1969-1974), the NaN mask (pf.py:1976-1981), temperature / dewpoint / height at 850, 700 and
//                   500 hPa (deep_convective_index pf.py:1830, lapse_rate pf.py:2102, isobar_temperature pf.py:2193: the
//                   log_interp rule of pf.py:1758-1828), the freezing level (pf.py:2137) and the melting level of the
//                   1/3-rule wet bulb (pf.py:2160, 364) -- what took a dewpoint pass, an interpolation pass, a wet-bulb
//                   array, two crossing passes and four isnan passes before, each re-reading the grid;
//   k_conv_finish   per point: mixing ratio of the most-unstable parcel (pf.py:2053-2059), the three deep convective
//                   indices, the 700-500 hPa lapse rate, the 0-6 km shear (pf.py:2216-2259) and the blanking of invalid
//                   points (pf.py:2097-2098).
// Per value the arithmetic is that of the stand-alone kernels (k_dewpoint_from_q, k_interp_levels, k_crossing_level).
#pragma once
#include "xp_kernels.hpp"
#include "xp_primitives.hpp"   // interp_column

namespace xp {

struct ConvColumnsArgs {
    View p, t, q, z;
    int64_t nlev, ncol;
    void *td;                   // (nlev, ncol) dense, element type T: the dewpoint the parcel passes read
    void *t850, *t700, *t500, *td850, *z700, *z500, *freezing, *melting;   // ncol each, element type T
    int32_t *valid;
};

// the 1/3-rule wet bulb in the arithmetic the array expression of pf.py:364-387 gets: the data's own type, one rounding per
// operation (no contraction)
template <typename T> XP_DEV double wet_bulb_third(double t, double td) {
#pragma clang fp contract(off)
    const T a = (T)t, b = (T)td;
    const T d = a - b;
    const T m = (T)(1.0 / 3.0) * d;
    return (double)(a - m);
}

template <typename T> __global__ __launch_bounds__(256) void k_conv_columns(ConvColumnsArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncol) return;
    constexpr int NT = 3, NV = 3;                          // targets 850, 700, 500 hPa; variables T, Td, z
    double at[NT] = {clog(850.0), clog(700.0), clog(500.0)};
    double cb[NT], ca[NT], sb[NV][NT], sa[NV][NT];
    int nb[NV][NT], na[NV][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        cb[j] = ca[j] = qnan();
#pragma unroll
        for (int v = 0; v < NV; ++v) { sb[v][j] = sa[v][j] = 0.0; nb[v][j] = na[v][j] = 0; }
    }
    bool valid = true;
    double best_t = qnan(), best_w = qnan(), x0 = qnan(), dt0 = qnan(), dw0 = qnan();
    T *const tdo = (T *)a.td;
    for (int64_t k = 0; k < a.nlev; ++k) {
        const double P = ld<T>(a.p, k, c), Tk = ld<T>(a.t, k, c), Q = ld<T>(a.q, k, c), Z = ld<T>(a.z, k, c);
        const T tdr = (T)dewpoint_from_q(P, Tk, Q);        // parcel_test.py:262-266, pf.py:1969: stored in the data's type ...
        tdo[k * a.ncol + c] = tdr;
        const double Td = (double)tdr;                     // ... and that stored value is what everything downstream reads
        valid = valid && !isnan_(P) && !isnan_(Tk) && !isnan_(Q) && !isnan_(Td);
        // linear_interp in ln p (pf.py:1758-1828), exactly k_interp_levels' bookkeeping
        const double cc = clog(P);
        double x[NV] = {Tk, Td, Z};
        if (!isnan_(cc)) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (cc >= at[j]) {
                    if (!(cc >= cb[j])) {
                        cb[j] = cc;
#pragma unroll
                        for (int v = 0; v < NV; ++v) { sb[v][j] = 0.0; nb[v][j] = 0; }
                    }
                    if (cc == cb[j]) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) if (!isnan_(x[v])) { sb[v][j] += x[v]; ++nb[v][j]; }
                    }
                }
                if (cc <= at[j]) {
                    if (!(cc <= ca[j])) {
                        ca[j] = cc;
#pragma unroll
                        for (int v = 0; v < NV; ++v) { sa[v][j] = 0.0; na[v][j] = 0; }
                    }
                    if (cc == ca[j]) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) if (!isnan_(x[v])) { sa[v][j] += x[v]; ++na[v][j]; }
                    }
                }
            }
        }
        // lowest crossing of 273.15 K in height (find_intersections pf.py:992-1064 + the min of pf.py:2153), k_crossing_level's rule
        const double dt1 = Tk - 273.15, dw1 = wet_bulb_third<T>(Tk, Td) - 273.15;
        if (k > 0) {
            {
                const double s0 = (double)((dt0 > 0.0) - (dt0 < 0.0)), s1 = (double)((dt1 > 0.0) - (dt1 < 0.0));
                if (isnan_(dt0) || isnan_(dt1) || s0 != s1) {
                    const double xi = (dt1 * x0 - dt0 * Z) / (dt1 - dt0);
                    if (!isnan_(xi) && !(xi >= best_t)) best_t = xi;
                }
            }
            {
                const double s0 = (double)((dw0 > 0.0) - (dw0 < 0.0)), s1 = (double)((dw1 > 0.0) - (dw1 < 0.0));
                if (isnan_(dw0) || isnan_(dw1) || s0 != s1) {
                    const double xi = (dw1 * x0 - dw0 * Z) / (dw1 - dw0);
                    if (!isnan_(xi) && !(xi >= best_w)) best_w = xi;
                }
            }
        }
        x0 = Z; dt0 = dt1; dw0 = dw1;
    }
    constexpr int f64 = sizeof(T) == 8;
    auto value = [&](int v, int j) __attribute__((always_inline)) {
        const double xb = nb[v][j] ? sb[v][j] / (double)nb[v][j] : qnan(), xa = na[v][j] ? sa[v][j] / (double)na[v][j] : qnan();
        const double res = xb + (xa - xb) * ((at[j] - cb[j]) / (ca[j] - cb[j]));
        return (xb == xa) ? xb : res;                       // pf.py:1802-1806
    };
    st(a.t850, f64, c, value(0, 0)); st(a.t700, f64, c, value(0, 1)); st(a.t500, f64, c, value(0, 2));
    st(a.td850, f64, c, value(1, 0)); st(a.z700, f64, c, value(2, 1)); st(a.z500, f64, c, value(2, 2));
    st(a.freezing, f64, c, best_t); st(a.melting, f64, c, best_w);
    a.valid[c] = valid ? 1 : 0;
}

struct ConvFinishArgs {
    int64_t ncol;
    int ignore_nans;
    // inputs (ncol each, element type T)
    const void *mu_p, *mu_td, *li[3], *t850, *t700, *t500, *td850, *z700, *z500, *hi_u, *hi_v, *sfc_u, *sfc_v;
    const int32_t *valid;
    // in / out: blanked in place where the point is invalid
    void *cape[3], *cin[3], *li_out[3], *freezing, *melting;
    // outputs
    void *mu_mixing_ratio, *dci[3], *lapse, *temp_500, *shear_u, *shear_v, *shear_mag;
    int32_t *positive_shear;
};

template <typename T> __global__ __launch_bounds__(256) void k_conv_finish(ConvFinishArgs a) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncol) return;
    constexpr int f64 = sizeof(T) == 8;
    const bool keep = a.ignore_nans || a.valid[c] != 0;     // out.where(valid_points) (pf.py:2097-2098)
    auto put = [&](void *p, double v) __attribute__((always_inline)) { st(p, f64, c, keep ? v : qnan()); };
    // mixing ratio of the most-unstable parcel: specific_humidity_from_dewpoint -> mixing_ratio_from_specific_humidity (pf.py:2053-2059)
    const double p = ld1<T>(a.mu_p, c), td = ld1<T>(a.mu_td, c);
    const double e = 6.112 * exp(17.67 * (td - 273.15) / (td - 29.65));
    const double w = EPS * e / (p - e), qs = w / (1.0 + w);
    put(a.mu_mixing_ratio, qs / (1.0 - qs));
    const double t850 = ld1<T>(a.t850, c), td850 = ld1<T>(a.td850, c), t700 = ld1<T>(a.t700, c), t500 = ld1<T>(a.t500, c);
    const double z700 = ld1<T>(a.z700, c), z500 = ld1<T>(a.z500, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double li = ld1<T>(a.li[i], c);
        put(a.dci[i], (t850 - 273.15) + (td850 - 273.15) - li);            // pf.py:1830 (Kunz 2009)
        put(a.li_out[i], li);
        put(a.cape[i], ld1<T>(a.cape[i], c)); put(a.cin[i], ld1<T>(a.cin[i], c));
    }
    put(a.lapse, (t500 - t700) / (z500 / 1000.0 - z700 / 1000.0));           // pf.py:2102
    put(a.temp_500, t500);
    put(a.freezing, ld1<T>(a.freezing, c)); put(a.melting, ld1<T>(a.melting, c));
    // wind_shear (pf.py:2216-2259)
    const double hu = ld1<T>(a.hi_u, c), hv = ld1<T>(a.hi_v, c), su = ld1<T>(a.sfc_u, c), sv = ld1<T>(a.sfc_v, c);
    const double du = hu - su, dv = hv - sv;
    put(a.shear_u, du); put(a.shear_v, dv); put(a.shear_mag, sqrt(du * du + dv * dv));
    a.positive_shear[c] = (keep && sqrt(hu * hu + hv * hv) > sqrt(su * su + sv * sv)) ? 1 : 0;
}


// ---- per-point products on top of the bundle -----------------------------------------------------------------------------
// significant_hail_parameter (pf.py:2261-2306, SPC SHIP) in the reference's operation order
XP_DEV double ship_value(double mucape, double mixing_ratio, double lapse, double temp_500, double shear, double flh) {
#pragma clang fp contract(off)
    mixing_ratio = mixing_ratio * 1e3;                                      // kg/kg -> g/kg
    lapse = -lapse;
    temp_500 = temp_500 - 273.15;
    shear = (shear >= 7.0 && shear <= 27.0) ? shear : qnan();               // validity windows (pf.py:2287-2289)
    mixing_ratio = (mixing_ratio >= 11.0 && mixing_ratio <= 13.6) ? mixing_ratio : qnan();
    temp_500 = (temp_500 <= -5.5) ? temp_500 : -5.5;
    double ship = mucape * mixing_ratio * lapse * -temp_500 * shear / 42000000.0;
    ship = (mucape >= 1300.0) ? ship : ship * (mucape / 1300.0);
    ship = (lapse >= 5.8) ? ship : ship * (lapse / 5.8);
    ship = (flh >= 2400.0) ? ship : ship * (flh / 2400.0);
    return ship;
}
template <typename T> __global__ __launch_bounds__(256)
void k_ship(int64_t n, const void *mucape, const void *mixing_ratio, const void *lapse, const void *temp_500, const void *shear,
            const void *flh, void *out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    st(out, sizeof(T) == 8, c, ship_value(ld1<T>(mucape, c), ld1<T>(mixing_ratio, c), ld1<T>(lapse, c), ld1<T>(temp_500, c),
                                          ld1<T>(shear, c), ld1<T>(flh, c)));
}

// storm_proxies (pf.py:2323-2407): the nine hail / storm proxies and SHIP from the bundle's per-point values.  Comparisons
// with NaN are false, as in NumPy.
struct ProxiesArgs {
    int64_t n;
    const void *mu_cape, *mu_mixing_ratio, *mixed_100_cape, *mixed_100_cin, *mixed_100_lifted_index, *mixed_100_dci,
               *mixed_50_cape, *mixed_50_cin, *lapse_rate_700_500, *temp_500, *freezing_level, *shear_magnitude;
    const int32_t *positive_shear;
    int32_t *proxy[9];   // Craven2004, Kunz2007, Trapp2007, Marsh2009, Allen2011, Allen2014, Eccel2012, Mohr2013, SHIP_0.1
    void *ship;
};
template <typename T> __global__ __launch_bounds__(256) void k_storm_proxies(ProxiesArgs a) {
#pragma clang fp contract(off)
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n) return;
    const double s06 = ld1<T>(a.shear_magnitude, c);
    double c100 = ld1<T>(a.mixed_100_cape, c), c50 = ld1<T>(a.mixed_50_cape, c), mucape = ld1<T>(a.mu_cape, c);
    c100 = (c100 >= 0.0) ? c100 : qnan(); c50 = (c50 >= 0.0) ? c50 : qnan(); mucape = (mucape >= 0.0) ? mucape : qnan();   // pf.py:2340-2343
    const double li100 = ld1<T>(a.mixed_100_lifted_index, c), dci100 = ld1<T>(a.mixed_100_dci, c);
    const double cin100 = ld1<T>(a.mixed_100_cin, c), cin50 = ld1<T>(a.mixed_50_cin, c), lapse = ld1<T>(a.lapse_rate_700_500, c);
    const bool pos = a.positive_shear[c] != 0;
    const double cs = c100 * s06;
    const bool allen11 = c50 * pow(s06, 1.67) >= 25000.0;
    const double ship = ship_value(mucape, ld1<T>(a.mu_mixing_ratio, c), lapse, ld1<T>(a.temp_500, c), s06, ld1<T>(a.freezing_level, c));
    const bool px[9] = {cs >= 20000.0,
                        (li100 <= -2.07) || (mucape >= 1474.0) || (dci100 >= 25.7),
                        (cs >= 10000.0) && (c100 >= 100.0) && (s06 >= 5.0) && pos,
                        cs >= 10000.0,
                        allen11,
                        allen11 && (cin50 > -25.0) && (s06 > 7.5) && (lapse < -6.5),
                        (cs > 10000.0) && (cin100 > -50.0),
                        (li100 <= -1.6) || (c100 >= 439.0) || (dci100 >= 26.4),
                        ship > 0.1};
#pragma unroll
    for (int i = 0; i < 9; ++i) sti(a.proxy[i], c, px[i] ? 1 : 0);
    st(a.ship, sizeof(T) == 8, c, ship);
}

// wind_shear (pf.py:2216-2259): the wind at `shear_height` (linear interpolation in height, pf.py:1758 rule) minus the surface wind
struct ShearArgs {
    View u, v, h;
    int64_t nwind, ncol;
    const void *sfc_u, *sfc_v;
    double shear_height;
    void *shear_u, *shear_v, *shear_mag;
    int32_t *positive_shear;
};
template <typename T> __global__ __launch_bounds__(256) void k_wind_shear(ShearArgs a) {
#pragma clang fp contract(off)
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncol) return;
    constexpr int f64 = sizeof(T) == 8;
    const double hu = interp_column<T>(a.h, a.u, a.nwind, c, a.shear_height, false);
    const double hv = interp_column<T>(a.h, a.v, a.nwind, c, a.shear_height, false);
    const double su = ld1<T>(a.sfc_u, c), sv = ld1<T>(a.sfc_v, c);
    const double du = hu - su, dv = hv - sv;
    st(a.shear_u, f64, c, du); st(a.shear_v, f64, c, dv); st(a.shear_mag, f64, c, sqrt(du * du + dv * dv));
    sti(a.positive_shear, c, (sqrt(hu * hu + hv * hv) > sqrt(su * su + sv * sv)) ? 1 : 0);
}

}  // namespace xp
