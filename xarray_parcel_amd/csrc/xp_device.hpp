// xp_device.hpp -- device-side building blocks of libxparcel (gfx950).
//
// One thread owns one atmospheric column and streams it bottom-up exactly once.
// The reference (traupach/xarray_parcel modules/parcel_functions.py, "pf.py") instead
// materialises an (N+1)-level profile with the LCL inserted and runs ~100 full-array
// passes over it; here the LCL is a *virtual node* emitted when the scan passes
// p_lcl, and everything lfc_el (pf.py:1066-1198) and cape_cin_base (pf.py:1291-1392)
// need is carried in registers as running extrema and prefix sums:
//
//   CAPE = Rd * (PosArea(up to EL) - PosArea(up to LFC)),  CIN = Rd * NegArea(up to LFC)
//
// where the LFC is either the first qualifying increasing crossing or the LCL
// (fall-backs of pf.py:1161-1185) and the EL is the last decreasing crossing or the
// column top -- both only known at the end, hence the snapshots.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xp {

// MetPy 1.4.1 constants (the reference's un-vendored dependency)
constexpr double RD = 287.04749097718457;
constexpr double EPS = 0.6219569100577033;
constexpr double KAPPA = 2.0 / 7.0;
constexpr double CP_D = RD / KAPPA;
constexpr double LV = 2.50084e6;
constexpr double VT_EPS = 0.608;      // hard-coded in pf.py:782
constexpr double RK4_H_MAX = 0.1;     // exact-mode step bound in ln p (shared with the oracle)
constexpr double LCL_SNAP = 1e-11;    // a level this close (relative) to p_lcl counts as lying on the LCL

#define XP_DEV __device__ __forceinline__

XP_DEV double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }
XP_DEV bool isnan_(double x) { return x != x; }

// ---- fp64 math without the special-case handling of the device library -----------------------------
// fp64 runs at half the fp32 VALU rate on CDNA4 and the library exp/log/pow/division carry ~2x the
// instructions these need here (arguments are finite and far from overflow; NaN still propagates).
XP_DEV double frcp(double x) {                       // ~1 ulp: v_rcp_f64 + two Newton steps
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
XP_DEV double fdiv(double a, double b) { return a * frcp(b); }
// a * b + c with the fp64 CONSTANT c as a scalar operand: one VALU instruction (v_fma_f64 v, v, v, s[..]) plus two s_mov on
// the scalar pipe.  Left to itself the compiler prefers v_fmac with the constant pre-loaded into the destination by two
// v_mov_b32 -- three VALU instructions per Horner step of a fixed-coefficient polynomial -- or, hoisting those moves out
// of the level loop, holds every coefficient in a VGPR pair through the loop.
XP_DEV double fma_sc(double a, double b, double c) {
#ifdef XP_NO_FMA_SC
    return __builtin_fma(a, b, c);
#endif
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
XP_DEV double fexp(double x) {                       // |rel err| < 3e-16 for |x| < 700
    double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = fma_sc(2.08767569878681e-09, r, 2.505210838544172e-08);                 // 1/12! r + 1/11!
    p = fma_sc(p, r, 2.755731922398589e-07);
    p = fma_sc(p, r, 2.7557319223985893e-06);
    p = fma_sc(p, r, 2.48015873015873e-05);
    p = fma_sc(p, r, 1.984126984126984e-04);
    p = fma_sc(p, r, 1.388888888888889e-03);
    p = fma_sc(p, r, 8.333333333333333e-03);
    p = fma_sc(p, r, 4.1666666666666664e-02);
    p = fma_sc(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)k);
}
XP_DEV double flog(double x) {                       // fdlibm e_log.c kernel, positive finite x (NaN -> NaN)
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);       // [0.5, 1)
    bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    double f = m - 1.0;
    double s = f * frcp(2.0 + f);
    double z = s * s, w = z * z;
    double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}
XP_DEV double fpow(double x, double y) { return fexp(y * flog(x)); }
// ln of a coordinate (interpolation in ln p, pf.py:1813): flog for positive finite arguments, the library's log -- with its
// -inf / NaN results -- for everything else (a divergent branch nobody takes on sane data).  ONE function for coordinates
// and targets, so that a level exactly on a target compares equal.
XP_DEV double clog(double x) {
    if (x > 0.0 && x < 1.0e300) return flog(x);
    return log(x);
}

// ---- thermodynamics ---------------------------------------------------------------------
// Bolton (1980): 6.112 fexp(17.67 (T-273.15)/(T-29.65)), written as fexp(17.67 - 17.67*243.5/(T-29.65))
XP_DEV double sat_vapor_pressure(double t) { return 6.112 * fexp(__builtin_fma(-4302.645, frcp(t - 29.65), 17.67)); }
// once per column (mixed-layer parcel): library log + IEEE division so that e = 0 -> -inf/inf -> NaN as in NumPy
XP_DEV double dewpoint_of_e(double e) { double v = log(e / 6.112); return 273.15 + 243.5 * v / (17.67 - v); }
XP_DEV double mix_of_e(double e, double p) { return EPS * fdiv(e, p - e); }
XP_DEV double sat_mix(double p, double t) { return mix_of_e(sat_vapor_pressure(t), p); }
XP_DEV double vapor_pressure(double p, double w) { return p * fdiv(w, EPS + w); }
// pf.py:684-710: RH(T, Td) * w_s(p, T) = [e_s(Td)/e_s(T)] eps e_s(T)/(p - e_s(T)) = eps e_s(Td)/(p - e_s(T))
XP_DEV double mixing_ratio(double t, double td, double p) {
    return EPS * fdiv(sat_vapor_pressure(td), p - sat_vapor_pressure(t));
}
XP_DEV double virt(double t, double w) { return t * (1.0 + VT_EPS * w); }
// Bolton (1980) eq. 39 (metpy.calc.equivalent_potential_temperature, pf.py:123)
XP_DEV double theta_e(double p, double t, double td) {
    double e = sat_vapor_pressure(td), r = mix_of_e(e, p);
    double tl = 56.0 + frcp(frcp(td - 56.0) + flog(fdiv(t, td)) * (1.0 / 800.0));
    double thl = t * fpow(fdiv(1000.0, p - e), KAPPA) * fpow(fdiv(t, tl), 0.28 * r);
    return thl * fexp(r * (1.0 + 0.448 * r) * (fdiv(3036.0, tl) - 1.78));
}

// out-of-line copies of the slow paths: called from rare, ballot-guarded branches so that their polynomial
// constants and temporaries do not occupy registers of the per-level loop
__device__ __attribute__((noinline)) double sat_vapor_pressure_slow(double t) { return sat_vapor_pressure(t); }
__device__ __attribute__((noinline)) double exp_slow(double x) { return exp(x); }
__device__ __attribute__((noinline)) double log_slow(double x) { return log(x); }
// x of an intersection exactly as the reference spells it (pf.py:1046), two roundings in the numerator (no FMA
// contraction) and an IEEE division: used where the result decides a "p* < p_lcl" tie
__device__ __attribute__((noinline)) double crossing_x_ref(double y, double yp, double X, double Xp) {
#pragma clang fp contract(off)
    double a = y * Xp;
    double b = yp * X;
    return (a - b) / (y - yp);
}

// One s_waitcnt lgkmcnt(0) for a group of LDS reads that has just been requested, pinned in place: left alone the compiler
// waits for every value separately, right before the instruction that consumes it (lgkmcnt(4), (3), (2) ... between the
// steps of a Horner evaluation) -- four or five more instructions per group, and an instruction of ANY kind costs a
// wavefront an issue slot (a wavefront issues at most one instruction per ~4-5 cycles; with four wavefronts per SIMD
// that, not the fp64 pipe, is what the kernel runs into: scripts/dbg/fma_lat.hip).  vmcnt / expcnt untouched.
XP_DEV void lds_wait_all() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- e_s(T) from an LDS-resident table ---------------------------------------------------------------
// The per-level path needs e_s six times per level (environment T and Td, three RK4 stages, the parcel).
// exp + reciprocal cost ~27 fp64 instructions each; instead every workgroup stages a table into LDS:
// 193 one-kelvin intervals over 137..330 K, a degree-5 polynomial in r = T - left edge each (Chebyshev
// interpolant of Bolton's formula built in long double by xp_init), stored coefficient-major so that the lanes of a
// wavefront -- whose temperatures fall in different intervals -- hit different LDS banks.  Out-of-range or NaN
// temperatures take the formula.  Relative error 1.5e-13 above 230 K, 1.6e-12 above 200 K, 9e-10 at 137 K (where e_s is
// 1e-7 hPa): below 1e-12 K in any virtual temperature, seven orders under the parity bar.  (Round 1 carried degree 7,
// 6e-14 everywhere: two more fp64 instructions and two more LDS reads per evaluation for digits nothing consumes -- 2 %
// of the family kernel, 5 % of the RK4 one, measured in a same-box A/B.)
constexpr double ES_T_LO = 137.0;
// Row stride 257 doubles: (a) more than the 255 x 8 B reach of ds_read2_b64 and not a multiple of 64, so every
// coefficient is its own ds_read_b64 (2 LDS cycles, banks (a/4) mod 64) instead of half a ds_read2_b64 (8 cycles per
// pair, banks mod 32: measured 48 % of all LDS cycles were bank conflicts with the merged reads); (b) odd, so that
// row c is rotated by c banks against row 0.  193 intervals (137 ... 330 K) leave 64 spare columns per row: rows 0 and 1
// carry the ln table there (1/c_i and ln c_i for 64 mantissa intervals), and the last row ends after its 193 entries
// -- 11.8 KB in all (degree 5), which together with the scan's LDS slots lets four workgroups share a CU.
#ifndef XP_ES_DEG
#define XP_ES_DEG 5
#endif
constexpr int ES_N = 193, ES_DEG = XP_ES_DEG, ES_STRIDE = 257, ES_TAB = ES_DEG * ES_STRIDE + ES_N;
constexpr int LOG_N = 64, LOG_OFF = ES_N;           // ln table: tb[LOG_OFF + i] = 1/c_i, tb[ES_STRIDE + LOG_OFF + i] = ln c_i
constexpr int LDS_TAB = ES_TAB;
// `all_in_range` is a wave-uniform promise by the caller that every lane's t lies inside the table (the per-level
// code tests T, Td and the parcel temperature once per level with margins, see in_table()); without it the
// range test is made here and out-of-table / NaN lanes take the formula.
XP_DEV double es_tab(const double *tb, double t, bool all_in_range = false) {
    double u = t - ES_T_LO;
    int i = (int)u;                                  // NaN -> 0
    bool ok = true;
    if (!all_in_range) {
        ok = (u >= 0.0) && (u < (double)ES_N);
        i = i < 0 ? 0 : (i > ES_N - 1 ? ES_N - 1 : i);
    }
    double r = __builtin_amdgcn_fract(u);            // the polynomials are in T - (left edge of the interval): no way back from the index
    const double *c = tb + i;
    // all coefficients requested, then ONE s_waitcnt lgkmcnt(0): left alone the compiler waits for each coefficient just
    // before its fma (lgkmcnt(4), (3), ... : five more instructions per evaluation, each of which takes an issue slot)
    static_assert(ES_DEG == 5, "es_tab is written out for degree 5");
    double k0 = c[0], k1 = c[1 * ES_STRIDE], k2 = c[2 * ES_STRIDE], k3 = c[3 * ES_STRIDE], k4 = c[4 * ES_STRIDE], k5 = c[5 * ES_STRIDE];
    lds_wait_all();
    double p = k5;
    p = __builtin_fma(p, r, k4);
    p = __builtin_fma(p, r, k3);
    p = __builtin_fma(p, r, k2);
    p = __builtin_fma(p, r, k1);
    p = __builtin_fma(p, r, k0);
    if (!all_in_range) {
        // behind a wave-uniform test and an asm barrier so that the compiler cannot fold the slow path into the
        // fast one as a select
        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
            if (!ok) {
                double tt = t;
                asm volatile("" : "+v"(tt));
                p = sat_vapor_pressure_slow(tt);
            }
        }
    }
    return p;
}
XP_DEV bool in_table(double t, double margin) { return (t >= ES_T_LO + margin) && (t < ES_T_LO + (double)ES_N - margin); }
// The same test with margin 0 for the per-level code, on the HIGH WORD of the double: both table edges (137 and 330) are
// integers, so their low words are zero and  lo <= t < hi  <=>  hi32(t) - hi32(lo) < hi32(hi) - hi32(lo)  as unsigned
// numbers (a negative or NaN t has a huge high word).  One subtraction per value, the values of a level combined with
// a max, one compare in all -- instead of two fp64 compares, their literals and the mask logic per value.
constexpr unsigned hi32_of(double x) { return (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32); }
constexpr unsigned ES_HI_LO = hi32_of(ES_T_LO), ES_HI_SPAN = hi32_of(ES_T_LO + (double)ES_N) - hi32_of(ES_T_LO);
static_assert((__builtin_bit_cast(unsigned long long, ES_T_LO) & 0xffffffffull) == 0 &&
              (__builtin_bit_cast(unsigned long long, ES_T_LO + (double)ES_N) & 0xffffffffull) == 0, "table edges must have zero low words");
XP_DEV unsigned table_dist(double t) { return (unsigned)__double2hiint(t) - ES_HI_LO; }     // < ES_HI_SPAN: inside the table
XP_DEV bool all_in_table(unsigned d) { return d < ES_HI_SPAN; }
XP_DEV unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
// ln(x) from the LDS table that follows the e_s table: x = 2^e m, m in [0.5,1) = c_i (1 + r), |r| < 2^-7;
// ln x = e ln2 + ln c_i + log1p(r), log1p by its series to r^7 (< 2e-18).  Positive finite x (NaN -> NaN).
// SHORT: log1p(r) through r^5 (|r| <= 2^-7: the next term, r^6 / 6, is 3.6e-14 -- 5e-15 of ln p).  For the ln p of the
// level loop: 3e-9 J/kg of CAPE, 4e-11 hPa of a crossing pressure at most, two fp64 instructions less per level (1.5 % of
// the family kernel, same-box A/B).  NOT for the q -> dewpoint chain, whose logarithm feeds a dewpoint that must compare
// equal to the temperature on saturated levels to the last bit.
template <bool SHORT = false> XP_DEV double log_tab(const double *tb, double x) {
    const double *lt = tb + LOG_OFF;
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);
    int i = (__double2hiint(x) >> 14) & (LOG_N - 1);       // the top six fraction bits = floor(128 m) - 64 (one v_bfe_u32)
    const double rc = lt[i], lc = lt[ES_STRIDE + i];
    lds_wait_all();
    double r = __builtin_fma(m, rc, -1.0);
    double q;
    if (SHORT) {
        q = fma_sc(0.2, r, -0.25);
    } else {
        q = fma_sc(1.0 / 7.0, r, -1.0 / 6.0);
        q = fma_sc(q, r, 0.2);
        q = fma_sc(q, r, -0.25);
    }
    q = fma_sc(q, r, 1.0 / 3.0);
    q = __builtin_fma(q, r, -0.5);
    q = __builtin_fma(q, r, 1.0);
    return __builtin_fma((double)e, 0.6931471805599453, __builtin_fma(q, r, lc));
}
// exp(u) for the dry adiabat of the level loops, T0 (p / p0)^kappa = T0 exp(kappa (ln p - ln p0)): |u| is a few tenths at most,
// so u = i / 64 + r with exp(i / 64) from 64 spare entries of the table block (row 2, i = -40 ... 23: u in [-0.63, 0.37]) and
// exp(r), |r| <= 1/128, by its series through r^5 (3e-16) -- ~20 instructions instead of the ~45 of fexp (two scalar moves per
// coefficient of its degree-12 polynomial).  u = 0 gives exactly 1 (the parcel's own level must reproduce its temperature bit
// for bit, pf.py:1117-1120).  A lane outside the range takes the library exp.
constexpr int EXPT_OFF = 2 * ES_STRIDE + ES_N, EXPT_LO = -40, EXPT_N = 64;
XP_DEV double dry_factor(const double *tb, double u) {
    const double k = __builtin_rint(u * 64.0);
    const bool ok = (k >= (double)EXPT_LO) && (k <= (double)(EXPT_LO + EXPT_N - 1));        // NaN: not ok
    const int i = ok ? (int)k - EXPT_LO : 0;
    const double e = tb[EXPT_OFF + i];
    const double r = __builtin_fma(k, -1.0 / 64.0, u);
    lds_wait_all();
    double q = fma_sc(1.0 / 120.0, r, 1.0 / 24.0);
    q = fma_sc(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5);
    q = __builtin_fma(q, r, 1.0);
    q = __builtin_fma(q, r, 1.0);
    double v = e * q;
    if (!ok) { double w = u; asm volatile("" : "+v"(w)); v = exp_slow(w); }                  // (per lane, rare, out of line)
    return v;
}
XP_DEV double mixing_ratio_tab(const double *tb, double t, double td, double p, bool fast = false) {
    return EPS * fdiv(es_tab(tb, td, fast), p - es_tab(tb, t, fast));
}
// The environment's virtual temperature of a level, T (1 + 0.608 w(T, Td, p)) (pf.py:839-843), for the level loops: the two
// constants as one, and the reciprocal with ONE Newton step on v_rcp_f64 (2.2e-15 relative, measured: scripts/dbg/rcp_err.hip;
// in Tv that is 1e-14 K, a fraction of its last bit) -- five instructions less per level than virt(t, mixing_ratio_tab(...)).
// Knife-edge nodes (saturated parcels) do not come through here: they take virt_ref, the reference's own operation order.
XP_DEV double frcp1(double x) {
    double y = __builtin_amdgcn_rcp(x);
    return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}
// (The dry parcel's virtual temperature is its temperature times the SAME factor at the parcel's own level: at that level
// parcel and environment are the same air, and the two virtual temperatures have to come out bit-identical -- their
// difference there decides "the parcel is never warmer than the environment", pf.py:1166-1169.)
// e_s at two temperatures at once, both promised to lie inside the table: the twelve coefficients are requested together,
// ONE wait, and the two Horner chains run interleaved (a dependent fp64 chain issues one instruction per ~8 cycles; two
// independent ones fill each other's gaps).  Same values as two es_tab calls, bit for bit.
XP_DEV void es_tab2(const double *tb, double t1, double t2, double &e1, double &e2) {
    const double u1 = t1 - ES_T_LO, u2 = t2 - ES_T_LO;
    const double *c1 = tb + (int)u1, *c2 = tb + (int)u2;
    const double r1 = __builtin_amdgcn_fract(u1), r2 = __builtin_amdgcn_fract(u2);
    static_assert(ES_DEG == 5, "es_tab2 is written out for degree 5");
    double a0 = c1[0], a1 = c1[1 * ES_STRIDE], a2 = c1[2 * ES_STRIDE], a3 = c1[3 * ES_STRIDE], a4 = c1[4 * ES_STRIDE], a5 = c1[5 * ES_STRIDE];
    double b0 = c2[0], b1 = c2[1 * ES_STRIDE], b2 = c2[2 * ES_STRIDE], b3 = c2[3 * ES_STRIDE], b4 = c2[4 * ES_STRIDE], b5 = c2[5 * ES_STRIDE];
    lds_wait_all();
    double p = __builtin_fma(a5, r1, a4), q = __builtin_fma(b5, r2, b4);
    p = __builtin_fma(p, r1, a3); q = __builtin_fma(q, r2, b3);
    p = __builtin_fma(p, r1, a2); q = __builtin_fma(q, r2, b2);
    p = __builtin_fma(p, r1, a1); q = __builtin_fma(q, r2, b1);
    e1 = __builtin_fma(p, r1, a0); e2 = __builtin_fma(q, r2, b0);
}
// ... and with a third independent chain: the Horner evaluation of the adiabat-family polynomial (c[8] ... c[0] at z)
XP_DEV void es_tab2_horner(const double *tb, double t1, double t2, const double *c, double z, double &e1, double &e2, double &h) {
    const double u1 = t1 - ES_T_LO, u2 = t2 - ES_T_LO;
    const double *c1 = tb + (int)u1, *c2 = tb + (int)u2;
    const double r1 = __builtin_amdgcn_fract(u1), r2 = __builtin_amdgcn_fract(u2);
    double a0 = c1[0], a1 = c1[1 * ES_STRIDE], a2 = c1[2 * ES_STRIDE], a3 = c1[3 * ES_STRIDE], a4 = c1[4 * ES_STRIDE], a5 = c1[5 * ES_STRIDE];
    double b0 = c2[0], b1 = c2[1 * ES_STRIDE], b2 = c2[2 * ES_STRIDE], b3 = c2[3 * ES_STRIDE], b4 = c2[4 * ES_STRIDE], b5 = c2[5 * ES_STRIDE];
    double v = __builtin_fma(c[8], z, c[7]);                                // (three steps of the long chain while the reads are in flight)
    v = __builtin_fma(v, z, c[6]);
    v = __builtin_fma(v, z, c[5]);
    lds_wait_all();
    double p = __builtin_fma(a5, r1, a4), q = __builtin_fma(b5, r2, b4);
    v = __builtin_fma(v, z, c[4]);
    p = __builtin_fma(p, r1, a3); q = __builtin_fma(q, r2, b3);
    v = __builtin_fma(v, z, c[3]);
    p = __builtin_fma(p, r1, a2); q = __builtin_fma(q, r2, b2);
    v = __builtin_fma(v, z, c[2]);
    p = __builtin_fma(p, r1, a1); q = __builtin_fma(q, r2, b1);
    v = __builtin_fma(v, z, c[1]);
    e1 = __builtin_fma(p, r1, a0); e2 = __builtin_fma(q, r2, b0);
    h = __builtin_fma(v, z, c[0]);
}
// PAIR: take es_tab2 on the promised-in-range path (twelve more VGPRs for a moment: the CAPE/CIN-only kernels have them, the
// all-outputs kernels of the searching parcels would spill)
template <bool PAIR = false> XP_DEV double virt_factor_tab(const double *tb, double t, double td, double p, bool fast) {
    double e_td, e_t;
    if (PAIR && fast) es_tab2(tb, td, t, e_td, e_t);
    else { e_td = es_tab(tb, td, fast); e_t = es_tab(tb, t, fast); }
    return __builtin_fma(e_td * frcp1(p - e_t), VT_EPS * EPS, 1.0);
}
template <bool PAIR = false> XP_DEV double virt_env_tab(const double *tb, double t, double td, double p, bool fast) {
    return t * virt_factor_tab<PAIR>(tb, t, td, p, fast);
}
// stage the table (global -> LDS); every thread of the block must call this before any early return
XP_DEV const double *stage_es_table(const double *g, double *lds) {
    for (int i = threadIdx.x; i < LDS_TAB; i += blockDim.x) lds[i] = g[i];
    __syncthreads();
    return lds;
}

// ln(theta_e): same ordering as theta_e (monotone), without the two pow and the exp -- what the most-unstable search
// actually needs (argmax over the layer, pf.py:127-128); ln from the LDS table
XP_DEV double ln_theta_e(const double *tb, double p, double t, double td) {
    double e = es_tab(tb, td), r = mix_of_e(e, p);
    double lt = log_tab(tb, t), ltd = log_tab(tb, td);
    double tl = 56.0 + frcp(frcp(td - 56.0) + (lt - ltd) * (1.0 / 800.0));
    double ltl = log_tab(tb, tl);
    return lt + KAPPA * (6.907755278982137 - log_tab(tb, p - e)) + 0.28 * r * (lt - ltl) +
           r * (1.0 + 0.448 * r) * (fdiv(3036.0, tl) - 1.78);
}

// the same chain on the per-level path of xp_cape_cin (XP_HUM_SPECIFIC): LDS tables and fast division;
// RH e_s(T) = w (p - e_s(T)) / eps.  q <= 0 or q >= 1 has no dewpoint (MetPy: log of a non-positive number -> NaN).
XP_DEV double dewpoint_from_q_tab(const double *tb, double p, double t, double q) {
    double e = fdiv(q, 1.0 - q) * (p - es_tab(tb, t)) * (1.0 / EPS);
    double v = log_tab(tb, e * (1.0 / 6.112));
    double td = 273.15 + 243.5 * fdiv(v, 17.67 - v);
    return (e > 0.0) ? td : qnan();
}

// ---- LCL: metpy.calc.lcl as a per-column Steffensen iteration (pf.py:609-682) -------------------
// Same iteration and stop rule as MetPy / the oracle, in the fast fp64 forms above: the result agrees with theirs to
// ~1e-13 relative, not to the last bit.  The LCL decides on which side of the condensation level every model level
// falls, and the reference's parcel virtual temperature jumps there by ~0.013 K (its w_parcel = RH * w_s(T) is not
// the w of this iteration), so a level within rounding of the LCL is a knife edge -- KAT
// test_profile_with_lcl_in_levels puts a level exactly on it.  The level loop therefore treats a level within 1e-11
// (relative) of p_lcl as lying ON the LCL (LCL_SNAP), which is what bitwise equality selects in the reference.
// (1e-11 = a hundred times the agreement of the two LCLs.  It was 1e-9 until late in round 3: a level between 1e-13 and the
// snap distance from the LCL, not on it, comes out on the other side than in the reference -- 4 of 17 M synthetic columns at
// 1e-9, scripts/run_gpu_soak_profile.py, each with CIN off by the ~0.1 J/kg of the virtual-temperature jump.)
XP_DEV double dewpoint_fast(double e) { double v = flog(e * (1.0 / 6.112)); return 273.15 + 243.5 * fdiv(v, 17.67 - v); }
XP_DEV double es_ref(double t) { return 6.112 * exp(17.67 * (t - 273.15) / (t - 29.65)); }
// metpy.calc.dewpoint_from_specific_humidity, MetPy 1.4.1 chain (parcel_test.py:262-266, pf.py:1889):
// w = q/(1-q); RH = w / w_s(p, T); Td = dewpoint(RH * e_s(T)) -- library exp/log, reference operation order
XP_DEV double dewpoint_from_q_ref(double p, double t, double q) {
    double w = q / (1.0 - q);
    double est = es_ref(t);
    double rh = w / (EPS * est / (p - est));
    return dewpoint_of_e(rh * est);
}
__device__ __attribute__((noinline)) double dewpoint_from_q_slow(double p, double t, double q) { return dewpoint_from_q_ref(p, t, q); }
// The same chain in the fast fp64 forms (RH e_s(T) = w (p - e_s(T)) / eps; ~1e-15 relative, 90 instead of 250 instructions) for
// physically sane arguments; anything else -- where IEEE infinities and the library's edge cases decide between a value and NaN --
// takes the reference spelling above, out of line behind a ballot.
XP_DEV double dewpoint_from_q(double p, double t, double q) {
    const bool sane = (t > 150.0) && (t < 350.0) && (p > 1.0) && (p < 2000.0) && (q > 0.0) && (q < 0.5);
    double e = fdiv(q, 1.0 - q) * (p - sat_vapor_pressure(t)) * (1.0 / EPS);
    double td = (e > 0.0) ? dewpoint_fast(e) : qnan();
    if (__builtin_amdgcn_ballot_w64(!sane) != 0ull && !sane) td = dewpoint_from_q_slow(p, t, q);
    return td;
}
XP_DEV double lcl_iter(double p, double p0, double w, double t) {
    double td = dewpoint_fast(p * fdiv(w, EPS + w));
    double r = fdiv(td, t);                              // (Td / T)^(1 / kappa) = r^3.5 = r^3 sqrt(r): ~15 instructions, not exp(3.5 ln r)
    return p0 * ((r * r) * (r * __builtin_sqrt(r)));
}
// pf.py:684-710 + 782-804 in the reference's own operation order (RH * w_s, then Tv), library math
XP_DEV double virt_ref(double t, double td, double p) {
    double est = es_ref(t);
    double w = (es_ref(td) / est) * (EPS * est / (p - est));
    return t * (1.0 + VT_EPS * w);
}
struct Lcl { double p, t, tv; int not_converged; };
// Library-math spelling (IEEE division, device-library exp/log/pow, the oracle's operation order): used for
// parcels outside the physically sane box, where the fixed point may diverge or leave the domain of log/pow and the
// outcome (NaN / not converged) has to follow IEEE semantics as on the CPU.  Out of line: rare.
__device__ __attribute__((noinline)) Lcl lcl_reference(double p_start, double t, double td) {
    Lcl r; r.not_converged = 0;
    double es_td = es_ref(td);
    double w = EPS * es_td / (p_start - es_td);
    double p0 = p_start, p = qnan();
    bool conv = false;
    for (int it = 0; it < 50; ++it) {
        double td1 = log(p0 * w / (EPS + w) / 6.112); td1 = 273.15 + 243.5 * td1 / (17.67 - td1);
        double p1 = p_start * pow(td1 / t, 1.0 / KAPPA);
        double td2 = log(p1 * w / (EPS + w) / 6.112); td2 = 273.15 + 243.5 * td2 / (17.67 - td2);
        double p2 = p_start * pow(td2 / t, 1.0 / KAPPA);
        double d = p2 - 2.0 * p1 + p0;
        p = (d != 0.0) ? p0 - (p1 - p0) * (p1 - p0) / d : p2;
        double rel = (p0 != 0.0) ? (p - p0) / p0 : p;
        if (fabs(rel) < 1e-5) { conv = true; break; }
        p0 = p;
    }
    if (!conv) { p = qnan(); r.not_converged = 1; }
    if (fabs(p - p_start) <= 1e-8 + 1e-5 * fabs(p_start)) p = p_start;
    r.p = p;
    double v = log(p * w / (EPS + w) / 6.112);
    r.t = 273.15 + 243.5 * v / (17.67 - v);
    r.tv = virt_ref(r.t, r.t, p);
    return r;
}
XP_DEV Lcl lcl(double p_start, double t, double td) {
    Lcl r;
    r.not_converged = 0;
    if (isnan_(p_start) || isnan_(t) || isnan_(td)) { r.p = r.t = r.tv = qnan(); return r; }   // pf.py:627-634, 680
    bool sane = (td <= t) && (td > 150.0) && (t < 350.0) && (p_start > 50.0) && (p_start < 1200.0);
    double es_td = sat_vapor_pressure(td);
    double w = EPS * fdiv(es_td, p_start - es_td);
    double p0 = p_start, p = qnan();
    bool conv = false;
    for (int it = 0; it < 50; ++it) {
        double p1 = lcl_iter(p0, p_start, w, t);
        double p2 = lcl_iter(p1, p_start, w, t);
        double d = p2 - 2.0 * p1 + p0;
        p = (d != 0.0) ? p0 - fdiv((p1 - p0) * (p1 - p0), d) : p2;
        double rel = (p0 != 0.0) ? fdiv(p - p0, p0) : p;
        if (fabs(rel) < 1e-5) { conv = true; break; }
        p0 = p;
        if (!sane) break;                                                  // handled below
    }
    if (!conv) { p = qnan(); r.not_converged = 1; }
    if (fabs(p - p_start) <= 1e-8 + 1e-5 * fabs(p_start)) p = p_start;    // np.isclose snap (MetPy issue #1187)
    r.p = p;
    r.t = dewpoint_fast(p * fdiv(w, EPS + w));
    // RH = 1 at the LCL (pf.py:653-657).  Fast forms (3e-16 relative; the library-math spelling cost ~150 instructions per
    // column); saturated parcels (LCL snapped onto the parcel level) take the reference spelling below, all of it: the sign
    // of T_lcl - T there is rounding noise of exactly those expressions (see the on-LCL handling in k_cape_cin)
    r.tv = virt(r.t, mix_of_e(sat_vapor_pressure(r.t), p));
    bool ref_path = !sane || (p == p_start);
    if (__builtin_amdgcn_ballot_w64(ref_path) != 0ull && ref_path) r = lcl_reference(p_start, t, td);
    return r;
}

// ---- moist adiabat ------------------------------------------------------------------------
// dT/dln p of MetPy's pseudo-adiabat, one division (same grouping as the oracle)
XP_DEV double dt_dlnp_e(double p, double t, double e) {
    double pe = p - e, rt2 = RD * t * t;
    double num = __builtin_fma(RD * t, pe, (LV * EPS) * e);
    double den = __builtin_fma(CP_D * rt2, pe, (LV * LV * EPS * EPS) * e);
    return rt2 * fdiv(num, den);
}
XP_DEV double dt_dlnp(const double *es, double p, double t, bool fast) { return dt_dlnp_e(p, t, es_tab(es, t, fast)); }

struct Tables {               // reference-format lookup tables resident in HBM (pf.py:447-523)
    const uint16_t *index;    // [n_p][n_t], 0 = NaN
    const float *adiabats;    // [n_ad][n_p] ascending pressure
    int64_t n_p, n_t;
    double p_max, p_step, t_min, t_step;
};

// Parcel temperature above the LCL.  Exact mode marches RK4 from node to node; table mode picks the
// adiabat row once (nearest neighbour, pf.py:554-556) and interpolates linearly in p (pf.py:585-592).
struct Moist {
    double x, p, t, e;        // exact mode: current point on the adiabat (ln p, p, T) and e_s(T) there
    const float *row;         // table mode: selected adiabat, nullptr = NaN
    const double *es;         // LDS e_s table
    bool table, dead;

    XP_DEV void start(const double *es_lds, double p_ref, double x_ref, double t_ref, bool table_mode, const Tables &tb) {
        es = es_lds; x = x_ref; p = p_ref; t = t_ref; table = table_mode; row = nullptr;
        e = es_tab(es, t_ref);
        dead = isnan_(p_ref) || isnan_(t_ref);
        if (table_mode && !dead) {
            double fi = (tb.p_max - p_ref) / tb.p_step, fj = (t_ref - tb.t_min) / tb.t_step;
            double i0 = floor(fi), j0 = floor(fj);
            // pandas nearest on a DEcreasing index keeps the left (higher-pressure) label on ties,
            // on an increasing index the right one
            int64_t ip = (int64_t)(((i0 + 1.0) - fi < fi - i0) ? i0 + 1.0 : i0);
            int64_t jt = (int64_t)(((j0 + 1.0) - fj <= fj - j0) ? j0 + 1.0 : j0);
            ip = ip < 0 ? 0 : (ip > tb.n_p - 1 ? tb.n_p - 1 : ip);
            jt = jt < 0 ? 0 : (jt > tb.n_t - 1 ? tb.n_t - 1 : jt);
            uint16_t a = tb.index[ip * tb.n_t + jt];
            if (a) row = tb.adiabats + (int64_t)(a - 1) * tb.n_p;
        }
    }
    // temperature of the adiabat at pressure pk (ln pk = xk); levels must come in order of
    // increasing distance from the start point (pressure decreasing upwards)
    // `hoist`: test the e_s table range once per RK4 step (wave-uniform) instead of once per stage: a step moves the
    // parcel temperature by at most kappa*T*h < 9 K, so 10 K of margin at its start covers all four stages
    XP_DEV double at(double pk, double xk, const Tables &tb, bool hoist = false) {
        if (dead || isnan_(pk)) return qnan();
        if (table) {
            // (table mode keeps IEEE division: it emulates np.interp on the reference's tables)
            double p_min = tb.p_max - (double)(tb.n_p - 1) * tb.p_step;
            if (row == nullptr || pk < p_min || pk > tb.p_max) return qnan();        // pf.py:598-600
            double f = (pk - p_min) / tb.p_step;
            int64_t lo = (int64_t)floor(f);
            if (lo >= tb.n_p - 1) return (double)row[tb.n_p - 1];
            double xlo = p_min + (double)lo * tb.p_step, xhi = p_min + (double)(lo + 1) * tb.p_step;
            double ylo = (double)row[lo], yhi = (double)row[lo + 1];
            return (yhi - ylo) / (xhi - xlo) * (pk - xlo) + ylo;                       // np.interp
        }
        if (xk != x) {
            double dx = xk - x;
            int ns = (int)ceil(fabs(dx) * (1.0 / RK4_H_MAX) - 1e-12);
            ns = ns < 1 ? 1 : ns;
            double h = dx * frcp((double)ns);
            double q = 0.5 * h;                                   // |q| <= 0.05: exp(q) by its Taylor series to q^8 (< 1e-17)
            double rh = 2.48015873015873e-05;
            rh = __builtin_fma(rh, q, 1.984126984126984e-04);
            rh = __builtin_fma(rh, q, 1.388888888888889e-03);
            rh = __builtin_fma(rh, q, 8.333333333333333e-03);
            rh = __builtin_fma(rh, q, 4.1666666666666664e-02);
            rh = __builtin_fma(rh, q, 1.6666666666666666e-01);
            rh = __builtin_fma(rh, q, 0.5);
            rh = __builtin_fma(rh, q, 1.0);
            rh = __builtin_fma(rh, q, 1.0);
            double ps = p;
            for (int s = 0; s < ns; ++s) {
                double pm = ps * rh;
                double pe = (s == ns - 1) ? pk : pm * rh;
                bool fast = hoist && (__builtin_amdgcn_ballot_w64(!in_table(t, 10.0)) == 0ull);
                double k1 = dt_dlnp_e(ps, t, e);                 // e_s(T) at the current point is already known
                double k2 = dt_dlnp(es, pm, t + 0.5 * h * k1, fast);
                double k3 = dt_dlnp(es, pm, t + 0.5 * h * k2, fast);
                double k4 = dt_dlnp(es, pe, t + h * k3, fast);
                t = t + (h * (1.0 / 6.0)) * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
                e = es_tab(es, t, fast);
                ps = pe;
            }
            x = xk; p = pk;
        }
        return t;
    }
};

// ---- "adiabat family" exact mode -----------------------------------------------------------------------------
// The one-parameter family of solutions of the pseudo-adiabat ODE, T(x ; psi) with x = ln p and psi = the adiabat's
// temperature at 1000 hPa, stored once -- as the parcel's VIRTUAL temperature along the adiabat,
// Tv = T (1 + 0.608 w_s(p, T)) (pf.py:760-775: what the CAPE / CIN integration consumes, so that the steady-state level
// needs no e_s of the parcel) -- in a piecewise polynomial (specification: oracle/family.py; built by xp_init):
//     x-pieces j < 8   : [XHI - 0.5 (j + 1), XHI - 0.5 j], XHI = ln 1100        (1100 ... ~20 hPa)
//     psi-pieces q < 9 : [EDGES[q], EDGES[q + 1]], 215 ... 312 K, narrower towards the warm end
//     Tv = sum_n sum_m A[j][n][m][q] z^n s^m,  z, s in [-1, 1] the piece-local coordinates,  n, m <= 8
// within 7e-7 K of the ODE (the RK4 stepper: 2e-5 K).  The parcel temperature, where it is wanted (profile output, no
// virtual-temperature correction), is the T that has this virtual temperature: Family::temperature_of.  A column finds its label once (one coarse RK4 march from the
// LCL to 1000 hPa, then three Newton steps on the table inside the psi-piece the coarse label falls in), collapses the
// psi direction of its current x-piece into nine coefficients held in registers, and from then on a level costs one
// Horner evaluation (~14 fp64 instructions instead of ~155 for an RK4 step) -- no memory access in the level loop.
// Moving into the next x-piece (every ~0.5 in ln p, i.e. 5-6 times per column) reloads the nine coefficients: 81
// reads whose addresses differ between the lanes of a wavefront only in q (9 adjacent doubles per (j, n, m): a broadcast
// plus adjacent banks when the table sits in LDS, one or two cache lines per instruction when it is read from L2 --
// whatever the lanes' labels are, unlike a per-lane table walk).
// Above the table top the adiabat continues dry (e_s / p < 1e-6 there).  A label or LCL outside the table marks the
// column `bad`; such columns are redone by the RK4 kernel.
constexpr double FAM_XHI = 7.003065458786462;    // ln 1100
constexpr double FAM_WX = 0.5, FAM_X1000 = 6.907755278982137;
constexpr int FAM_NPX = 8, FAM_ND = 8, FAM_MD = 8, FAM_NPS = 9;
constexpr double FAM_XLO = FAM_XHI - FAM_WX * FAM_NPX;
constexpr int FAM_COEFS = FAM_NPX * (FAM_ND + 1) * (FAM_MD + 1) * FAM_NPS;
constexpr int FAM_SIZE = FAM_COEFS + 2 * FAM_NPS;         // + psi-piece centres and 1 / half-widths (appended by the host)
constexpr double FAM_LABEL_H = 0.25, FAM_MARGIN = 0.05;
constexpr double FAM_PSI_LO = 215.0, FAM_PSI_HI = 312.0;
#define XP_FAM_EDGES {215.0, 245.0, 262.0, 275.0, 285.0, 293.0, 299.0, 304.0, 308.5, 312.0}

// device layout of the table: A[j][n][m][q] sits at ((m * 9 + n) * 8 + j) * 9 + q (see family_device_layout, xparcel.hip)
constexpr int FAM_SM = (FAM_ND + 1) * FAM_NPX * FAM_NPS, FAM_SN = FAM_NPX * FAM_NPS, FAM_SJ = FAM_NPS;
struct Family {
    const double *tab;     // coefficient table in the device layout, then centre[q], 1 / half-width[q]
    double c[FAM_ND + 1];  // the column's polynomial in z on x-piece jx
    double m4;             // z = 4 x + m4
    double s;              // the label's coordinate in psi-piece q
    int q;
    bool bad;

    XP_DEV static int psi_piece(double psi) {
        return (int)(psi >= 245.0) + (int)(psi >= 262.0) + (int)(psi >= 275.0) + (int)(psi >= 285.0) + (int)(psi >= 293.0) +
               (int)(psi >= 299.0) + (int)(psi >= 304.0) + (int)(psi >= 308.5);
    }
    XP_DEV static double x_mid(int j) { return FAM_XHI - FAM_WX * ((double)j + 0.5); }
    // c_n = sum_m A[j][n][m][q] s^m: nine reads in flight at a time
    XP_DEV void load_piece(int j) {
        const double *a = tab + (j * FAM_SJ + q);
#pragma unroll
        for (int n = 0; n <= FAM_ND; ++n) {
            const double *r = a + n * FAM_SN;
            double k[FAM_MD + 1];
#pragma unroll
            for (int m = 0; m <= FAM_MD; ++m) k[m] = r[m * FAM_SM];
            lds_wait_all();                              // nine reads in flight, one wait
            double v = k[FAM_MD];
#pragma unroll
            for (int m = FAM_MD - 1; m >= 0; --m) v = __builtin_fma(v, s, k[m]);
            asm volatile("" : "+v"(v) : : "memory");     // this row is finished before the next row's reads are issued
            c[n] = v;
        }
        m4 = -(2.0 / FAM_WX) * x_mid(j);
    }
    // the T with T (1 + 0.608 w_s(p, T)) = tv: five Newton steps from T0 = tv / (1 + 0.608 w_s(p, tv)) (1e-13 K wherever
    // e_s <= 0.1 p; oracle/family.py temperature_of).  Only the kernels that output or integrate the plain temperature pay it.
    XP_DEV static double temperature_of(const double *es, double p, double tv) {
        constexpr double c = VT_EPS * EPS;
        double e0 = es_tab(es, tv);
        double t = fdiv(tv, 1.0 + c * fdiv(e0, p - e0));
#pragma nounroll
        for (int it = 0; it < 5; ++it) {
            double e = es_tab(es, t);
            double rt = frcp(t - 29.65), rp = frcp(p - e);
            double de = e * (17.67 * 243.5) * (rt * rt);
            double g = c * e * rp;
            double f = __builtin_fma(t, g, t) - tv;
            double df = 1.0 + g + t * c * p * de * (rp * rp);
            t = t - fdiv(f, df);
        }
        return t;
    }
    // The same root from a WARM start: `off` = Tv - T of the node before (it changes by a few hundredths of a kelvin per
    // level; the LCL's own Tv - T to begin with).  Three Newton steps reach the 1e-13 K of the five cold ones
    // (5.7e-14 on 50-level columns); a residual test sends coarse columns -- where the offset moved too far -- through two
    // more.  NaN in (missing pressure) leaves `off` alone.
    // Only the residual f has to be exact: the slope enters the step alone, so its two reciprocals are the bare v_rcp_f64
    // (2^-23 relative: the step after a residual of 1e-6 K still lands within 1.2e-13 K) -- 14 instructions less per step.
    // `all_in_range`: wave-uniform promise that every lane's tv - off lies inside the e_s table (see es_tab).
    XP_DEV static double temperature_from(const double *es, double p, double tv, double &off, bool all_in_range = false) {
        constexpr double c = VT_EPS * EPS;
        double t = tv - off, f = 0.0;
#pragma nounroll
        for (int it = 0; it < 3; ++it) {
            double e = es_tab(es, t, all_in_range);
#ifdef XP_NEWTON_EXACT_SLOPE
            double rt = frcp(t - 29.65), rp = frcp(p - e);
#else
            double rt = __builtin_amdgcn_rcp(t - 29.65), rp = frcp(p - e);
#endif
            double de = e * (17.67 * 243.5) * (rt * rt);
            double g = c * e * rp;
            f = __builtin_fma(t, g, t) - tv;
            double df = 1.0 + g + t * c * p * de * (rp * rp);
#ifdef XP_NEWTON_EXACT_SLOPE
            t = t - fdiv(f, df);
#else
            t = t - f * __builtin_amdgcn_rcp(df);
#endif
        }
        // f is the residual BEFORE the last step; quadratic convergence: the step after a residual below 1e-6 K lands within 1e-13
        bool more = !(fabs(f) <= 1e-6);
        if (__builtin_amdgcn_ballot_w64(more && !isnan_(tv)) != 0ull) {
#pragma nounroll
            for (int it = 0; it < 2; ++it) {
                double e = es_tab(es, t);
                double rt = frcp(t - 29.65), rp = frcp(p - e);
                double de = e * (17.67 * 243.5) * (rt * rt);
                double g = c * e * rp;
                double f2 = __builtin_fma(t, g, t) - tv;
                double df = 1.0 + g + t * c * p * de * (rp * rp);
                t = more ? t - fdiv(f2, df) : t;
            }
        }
        if (!isnan_(t)) off = tv - t;
        return t;
    }
    // The same inversion for the level loop above the LCLs: the offset Tv - T is extrapolated from the two nodes before (it falls
    // smoothly with height: second differences of ~0.02 K near the ground, less above), so TWO Newton steps do -- a start within
    // 0.04 K leaves 2e-5 K after the first step (|f''/2f'| ~ 0.012 / K) and 5e-12 K after the second; a residual above 3e-5 K before
    // the last step (the first moist node of a column, a jump in the level spacing) sends the lane through two exact steps more.
    XP_DEV static double temperature_from2(const double *es, double p, double tv, double &off, double &offp, bool all_in_range) {
        constexpr double c = VT_EPS * EPS;
        double t = tv - (2.0 * off - offp), f = 0.0;
#pragma nounroll
        for (int it = 0; it < 2; ++it) {
            double e = es_tab(es, t, all_in_range);
            double rt = __builtin_amdgcn_rcp(t - 29.65), rp = frcp(p - e);
            double de = e * (17.67 * 243.5) * (rt * rt);
            double g = c * e * rp;
            f = __builtin_fma(t, g, t) - tv;
            double df = 1.0 + g + t * c * p * de * (rp * rp);
            t = t - f * __builtin_amdgcn_rcp(df);
        }
        bool more = !(fabs(f) <= 3e-5);
        if (__builtin_amdgcn_ballot_w64(more && !isnan_(tv)) != 0ull) {
#pragma nounroll
            for (int it = 0; it < 2; ++it) {
                double e = es_tab(es, t);
                double rt = frcp(t - 29.65), rp = frcp(p - e);
                double de = e * (17.67 * 243.5) * (rt * rt);
                double g = c * e * rp;
                double f2 = __builtin_fma(t, g, t) - tv;
                double df = 1.0 + g + t * c * p * de * (rp * rp);
                t = more ? t - fdiv(f2, df) : t;
            }
        }
        if (!isnan_(t)) { offp = off; off = tv - t; }
        return t;
    }
    XP_DEV double horner(double z) const {
        double v = c[FAM_ND];
#pragma unroll
        for (int n = FAM_ND - 1; n >= 0; --n) v = __builtin_fma(v, z, c[n]);
        return v;
    }
    // label of the adiabat through the LCL (x_lcl = ln p_lcl, t_lcl; tv_lcl its virtual temperature there) and the
    // coefficients of the x-piece the LCL is in
    XP_DEV void start(const double *table, const double *es, double p_lcl, double x_lcl, double t_lcl, double tv_lcl) {
        tab = table; q = 0; s = 0.0; m4 = 0.0;
        bad = !(x_lcl >= FAM_XLO && x_lcl <= FAM_XHI) || isnan_(t_lcl);
        // coarse label: RK4 to ln 1000 in steps <= 0.25
        double dx = FAM_X1000 - x_lcl;
        int ns = (int)ceil(fabs(dx) * (1.0 / FAM_LABEL_H) - 1e-12);
        ns = ns < 1 ? 1 : ns;
        ns = bad ? 1 : ns;
        double h = bad ? 0.0 : dx * frcp((double)ns);
        double qh = 0.5 * h;                                  // |qh| <= 0.125: exp by its Taylor series to qh^10 (< 1e-17)
        double rh = 2.755731922398589e-07;
        rh = __builtin_fma(rh, qh, 2.7557319223985893e-06);
        rh = __builtin_fma(rh, qh, 2.48015873015873e-05);
        rh = __builtin_fma(rh, qh, 1.984126984126984e-04);
        rh = __builtin_fma(rh, qh, 1.388888888888889e-03);
        rh = __builtin_fma(rh, qh, 8.333333333333333e-03);
        rh = __builtin_fma(rh, qh, 4.1666666666666664e-02);
        rh = __builtin_fma(rh, qh, 1.6666666666666666e-01);
        rh = __builtin_fma(rh, qh, 0.5);
        rh = __builtin_fma(rh, qh, 1.0);
        rh = __builtin_fma(rh, qh, 1.0);
        double t = t_lcl, ps = p_lcl;
#pragma nounroll
        for (int k = 0; k < ns; ++k) {
            double pm = ps * rh, pe = pm * rh;
            double k1 = dt_dlnp(es, ps, t, false);
            double k2 = dt_dlnp(es, pm, t + 0.5 * h * k1, false);
            double k3 = dt_dlnp(es, pm, t + 0.5 * h * k2, false);
            double k4 = dt_dlnp(es, pe, t + h * k3, false);
            t = t + (h * (1.0 / 6.0)) * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
            ps = pe;
        }
        const double psi0 = t;
        if (!(psi0 >= FAM_PSI_LO + FAM_MARGIN && psi0 <= FAM_PSI_HI - FAM_MARGIN)) bad = true;
        q = bad ? 0 : psi_piece(psi0);
        const double mid = tab[FAM_COEFS + q], inv_h = tab[FAM_COEFS + FAM_NPS + q];
        double u = __builtin_fma(-(1.0 / FAM_WX), x_lcl, FAM_XHI * (1.0 / FAM_WX));
        int j0 = bad ? 0 : (int)u;
        j0 = j0 > FAM_NPX - 1 ? FAM_NPX - 1 : j0;
        // Newton on the table inside piece q: T(x_lcl ; s) = sum_m b_m s^m with b_m = sum_n A[j0][n][m][q] z^n (one pass
        // over the 81 coefficients), value and derivative by one Horner sweep per step
        const double z = bad ? 0.0 : (x_lcl - x_mid(j0)) * (2.0 / FAM_WX);
        const double *a = tab + (j0 * FAM_SJ + q);
        double b[FAM_MD + 1];
#pragma unroll
        for (int m = 0; m <= FAM_MD; ++m) {
            double k[FAM_ND + 1];
#pragma unroll
            for (int n = 0; n <= FAM_ND; ++n) k[n] = a[m * FAM_SM + n * FAM_SN];
            lds_wait_all();
            double v = k[FAM_ND];
#pragma unroll
            for (int n = FAM_ND - 1; n >= 0; --n) v = __builtin_fma(v, z, k[n]);
            asm volatile("" : "+v"(v) : : "memory");
            b[m] = v;
        }
        double psi = psi0;
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            double sc = (psi - mid) * inv_h;
            double val = 0.0, der = 0.0;
#pragma unroll
            for (int m = FAM_MD; m >= 0; --m) {
                der = __builtin_fma(der, sc, val);
                val = __builtin_fma(val, sc, b[m]);
            }
            psi = psi - fdiv(val - tv_lcl, der * inv_h);
        }
        if (!(fabs(psi - psi0) <= FAM_MARGIN)) bad = true;
        s = bad ? 0.0 : (psi - mid) * inv_h;
        load_piece(j0);
        if (bad) poison();
    }
    // VIRTUAL temperature of the column's parcel at ln p = X (X <= x_lcl; levels normally come with decreasing X).
    // x-piece j holds z in (-1, 1]: the fast path only tests that (the piece index itself is recovered from m4 when a
    // lane has to move, with the oracle's floor rule).
    // at() in two steps, for a caller that has other arithmetic to interleave with the Horner chain (eight dependent fp64 fma):
    // locate() makes sure the coefficients of X's piece are loaded and returns the piece-local coordinate; value() evaluates.
    // any_top (wave-uniform): some lane is above the table top -- only then value() looks at the per-lane `top`.
    XP_DEV double locate(double X, bool &top, bool &any_top) {
        double z = __builtin_fma(2.0 / FAM_WX, X, m4);
        top = false; any_top = false;
        bool move = (z <= -1.0) || (z > 1.0);
        if (__builtin_amdgcn_ballot_w64(move) != 0ull) {
            double u = __builtin_fma(-(1.0 / FAM_WX), X, FAM_XHI * (1.0 / FAM_WX));
            bool go = move;
            if (go && u < 0.0) { bad = true; go = false; }
            int jn = (int)u;
            top = go && jn > FAM_NPX - 1;
            int jcur = (int)__builtin_rint(__builtin_fma(m4, 0.5, FAM_XHI * (1.0 / FAM_WX) - 0.5));
            int j = go ? (jn > FAM_NPX - 1 ? FAM_NPX - 1 : jn) : jcur;
            load_piece(bad ? 0 : j);
            if (bad) poison();
            z = __builtin_fma(2.0 / FAM_WX, X, m4);
            any_top = __builtin_amdgcn_ballot_w64(top) != 0ull;
        }
        return z;
    }
    XP_DEV double top_value(double v, double X, bool top) const {           // (rare: the dry continuation above the table top)
        const double w = horner(-1.0);
        return top ? w * fexp(KAPPA * (X - FAM_XLO)) : v;
    }
    XP_DEV double at(double X) {
        double z = __builtin_fma(2.0 / FAM_WX, X, m4);
        // a NaN z stays put: a NaN pressure, or a column outside the table (poisoned below: its nine coefficients and m4
        // are NaN, so it evaluates to NaN without a test of its own on the per-level path)
        bool move = (z <= -1.0) || (z > 1.0);
        if (__builtin_amdgcn_ballot_w64(move) != 0ull) {                            // rare: another x-piece, or off the table
            // every lane of the wavefront reloads -- the ones that stay put their current piece -- so that the nine
            // coefficients are plainly overwritten instead of merged per lane (which would keep old and new alive together)
            double u = __builtin_fma(-(1.0 / FAM_WX), X, FAM_XHI * (1.0 / FAM_WX)); // (XHI - X) / WX
            bool go = move;
            if (go && u < 0.0) { bad = true; go = false; }                          // p > 1100 hPa
            int jn = (int)u;
            bool top = go && jn > FAM_NPX - 1;                                      // above the table top: dry continuation
            int jcur = (int)__builtin_rint(__builtin_fma(m4, 0.5, FAM_XHI * (1.0 / FAM_WX) - 0.5));
            int j = go ? (jn > FAM_NPX - 1 ? FAM_NPX - 1 : jn) : jcur;
            load_piece(bad ? 0 : j);
            if (bad) poison();
            z = __builtin_fma(2.0 / FAM_WX, X, m4);
            if (__builtin_amdgcn_ballot_w64(top) != 0ull) {
                double v = horner(top ? -1.0 : z);
                return top ? v * fexp(KAPPA * (X - FAM_XLO)) : v;
            }
        }
        return horner(z);
    }
    XP_DEV void poison() {
#pragma unroll
        for (int n = 0; n <= FAM_ND; ++n) c[n] = qnan();
        m4 = qnan();
    }
};

// ---- the streaming LFC / EL / CAPE / CIN state machine ------------------------------------------
// Nodes are the levels of the LCL-augmented profile in order; feed them with node().  The per-node path is
// branch-free (selects); everything that happens at most a few times per column -- first node, sign changes,
// NaN gaps -- sits behind one ballot-guarded branch.
// Input contract used here (the reference's, README.md:9 / pf.py:2319-2320): pressure decreases with the node
// index, so "the lowest pressure where both temperatures exist" (pf.py:1143-1147) is the LAST such node.
// Values that are written a few times per column (at the LCL node, at sign changes) and read once (in finish) live in
// LDS, one slot per thread, instead of occupying VGPRs through the level loop: field f of thread t at slot[f * 256 + t].
#ifndef XP_CAPE_THREADS
#define XP_CAPE_THREADS 256
#endif
#ifndef XP_SLOT_FIELDS
#define XP_SLOT_FIELDS (XP_CAPE_THREADS >= 1024 ? 12 : 13)    // (the 1024-thread family workgroups have no LDS left for SL_LI)
#endif
constexpr int SLOT_STRIDE = XP_CAPE_THREADS, SLOT_FIELDS = XP_SLOT_FIELDS;
// SL_A0..A3 are used twice: below the LCL they hold the lower bracket of the LCL interpolation (kernel side:
// pressure, ln p, T, Td of the last valid level), from the LCL node on the bottom-LFC record -- an LFC has to lie
// above the LCL (pf.py:1127-1132), so the two never coexist; the LCL node re-initialises them.
enum { SL_CAPE_LCL = 0, SL_CIN_LCL, SL_MIN_P, SL_IDX /* two int32: lfc index, el index */, SL_LCL_T,
       SL_A0, SL_A1, SL_A2, SL_A3, SL_EL_X, SL_EL_T, SL_CAPE_EL, SL_LI /* profile kernels: lifted-index state */,
       SL_BR_P = SL_A0, SL_BR_X = SL_A1, SL_BR_T = SL_A2, SL_BR_TD = SL_A3,
       SL_LFC_X = SL_A0, SL_LFC_T = SL_A1, SL_CAPE_LFC = SL_A2, SL_CIN_LFC = SL_A3 };

struct Scan {
    // configuration
    double p_lcl, x_lcl;   // LCL pressure and its logarithm (the X of the LCL node)
    bool pos_neg;
    double *slot;          // this thread's LDS slots (SLOT_FIELDS doubles, stride SLOT_STRIDE)
    // previous node
    double Xp, yp, parp;
    int j;                 // nodes seen so far
    bool use_all;          // env[0] != par[0] (pf.py:1117-1120)
    // prefix sums (their snapshots at the LCL node / LFC / EL are in the slots)
    double cape, cin;
    // crossings, kept as ln p: pressures decrease along the scan, so "bottom LFC" / "top EL" order the same in ln p,
    // and the two exponentials a column actually needs are taken once, in finish()
    int any_inc;             // (an int, not a bool: set inside divergent code, a mask in scalar registers would have to be merged back at every join of every node)
    bool pos_parcel, env_any;
    bool top_le, any_valid;  // at the last node where p, parcel, environment all exist: parcel <= environment; there is one
    bool bad_p;              // a pressure higher than the node before it (outside the input contract)

    XP_DEV void init(double p_lcl_, double x_lcl_, bool pos_neg_, double *slot_) {
        p_lcl = p_lcl_; x_lcl = x_lcl_; pos_neg = pos_neg_; slot = slot_;
        Xp = yp = parp = qnan(); j = 0; use_all = true;
        cape = cin = 0.0;
        for (int f = 0; f < SLOT_FIELDS; ++f)
            slot[f * SLOT_STRIDE] = (f == SL_LFC_T || f == SL_EL_T || f == SL_LFC_X || f == SL_EL_X || f == SL_MIN_P || f == SL_LI) ? qnan() : 0.0;
        idx()[0] = -1; idx()[1] = -1;
        any_inc = 0;
        pos_parcel = env_any = top_le = any_valid = bad_p = false;
    }
    XP_DEV int *idx() const { return (int *)(slot + SL_IDX * SLOT_STRIDE); }     // [0] lfc index, [1] el index
    XP_DEV void add(double a) {                      // skip-NaN sums (pf.py:206) with the sign filters of pf.py:201-204
        if (pos_neg) { cape += fmax(a, 0.0); cin += fmin(a, 0.0); }      // maxNum/minNum drop a NaN operand
        else { double b = isnan_(a) ? 0.0 : a; cape += b; cin += b; }
    }
    // pressure of a crossing stored as ln p; one within 2e-9 of the LCL takes the library exp, like the CPU
    XP_DEV double crossing_pressure(double xs) const {
        double ps = fexp(xs);
        bool near_lcl = fabs(xs - x_lcl) <= 2e-9;
        if (__builtin_amdgcn_ballot_w64(near_lcl) != 0ull && near_lcl) {
            double q = xs;
            asm volatile("" : "+v"(q));
            ps = exp_slow(q);
        }
        return ps;
    }
    // first node, or an interval whose end points differ in sign / are NaN (pf.py:1019-1022)
    // ABOVE: the caller guarantees that this node and the one before it lie strictly above the LCL (the kernel's phase B):
    // every crossing then is "above the LCL" and none can tie with it
    template <bool LEAN, bool ABOVE> XP_DEV void special(double X, double par, double env, double y, double a_reg) {
        if (!ABOVE && j == 0) { use_all = (env != par); return; }
        int i = j - 1;
        // Some lane of a wavefront has a crossing in most iterations when neighbouring columns are unrelated (the
        // bench's are), so this path is not rare per wave: one fast reciprocal, no exp / ln.  Exceptions, behind
        // ballots: the zero-width interval of a duplicated pressure (IEEE 0/0 must give NaN as in NumPy), and a
        // crossing within 2e-9 (in ln p) of the LCL, whose "p* < p_lcl" tie is broken with the library exp / log
        // exactly as on the CPU.
        double d = y - yp, xs, frac;
        if constexpr (ABOVE) {
            // Strictly above the LCL (the kernel's phase B, most of a column): no duplicated pressure can occur there (the
            // only one a profile has is the LCL node on a level), and ONE test covers every "no valid zero" case -- y, yp, X
            // or Xp missing makes xs NaN, and then both triangles and the plain trapezoid are NaN too: nothing to add,
            // nothing to record.  (Twelve instructions less than the general form below on a path that some lane of a
            // wavefront takes at ~90 % of the levels of the bench's columns.)
            // (frcp1: 2.2e-15 relative, 1.5e-14 in the crossing's ln p.  The general form below uses the same reciprocal: whether a
            // node is fed in phase A or in phase B depends on the OTHER columns of the wavefront, so the two forms have to agree
            // to the last bit -- tests/test_gpu_parity.py::test_full_size_properties_config2, permutation equivariance)
            const double r = frcp1(d);
            xs = (y * Xp - yp * X) * r;
            if (isnan_(xs)) return;
            frac = -yp * r;
            add((yp * 0.5) * fabs(Xp - xs));                                // lower triangle (pf.py:1246-1273)
            if (y > 0.0) {                                                  // increasing crossing
                any_inc = 1;
                if (!(xs <= slot[SL_LFC_X * SLOT_STRIDE])) {                // bottom LFC (pf.py:1127-1132)
                    slot[SL_LFC_X * SLOT_STRIDE] = xs; if (!LEAN) idx()[0] = j - 1;
                    if (!LEAN) slot[SL_LFC_T * SLOT_STRIDE] = frac * (par - parp) + parp;     // pf.py:1050
                    slot[SL_CAPE_LFC * SLOT_STRIDE] = cape; slot[SL_CIN_LFC * SLOT_STRIDE] = cin;
                }
            }
            if (y < 0.0 && !(xs >= slot[SL_EL_X * SLOT_STRIDE])) {          // top EL (pf.py:1136-1138)
                slot[SL_EL_X * SLOT_STRIDE] = xs; if (!LEAN) idx()[1] = j - 1;
                if (!LEAN) slot[SL_EL_T * SLOT_STRIDE] = frac * (par - parp) + parp;
                slot[SL_CAPE_EL * SLOT_STRIDE] = cape;
            }
            add((y * 0.5) * fabs(X - xs));                                  // upper triangle
            return;
        }
        bool dup = (X == Xp);
        if (__builtin_amdgcn_ballot_w64(dup) != 0ull && dup) {
            xs = (y * Xp - yp * X) / d;                                     // pf.py:1046
            frac = (xs - Xp) / (X - Xp);
        } else {
            double r = frcp1(d);                                            // (the same reciprocal as the ABOVE form: see there)
            xs = (y * Xp - yp * X) * r;
            frac = -yp * r;                                                 // = (xs - Xp) / (X - Xp)
        }
        double zy = frac * d + yp;                                          // zero crossing of y (pf.py:1225-1231)
        if (isnan_(zy)) { add(a_reg); return; }                             // no valid zero: plain trapezoid (NaN -> skipped)
        double zlog = xs;                                                   // ln(exp(xs)) (pf.py:1237) = xs to 1 ulp
        bool above = ABOVE || xs < x_lcl;                                   // p* < p_lcl
        bool near_lcl = !ABOVE && fabs(xs - x_lcl) <= 2e-9;
        if (!ABOVE && __builtin_amdgcn_ballot_w64(near_lcl) != 0ull && near_lcl) {
            double q = y;
            asm volatile("" : "+v"(q));
            xs = crossing_x_ref(q, yp, X, Xp);                              // the tie is decided on the reference's own rounding
            double ps = exp_slow(xs);
            zlog = log_slow(ps);
            above = ps < p_lcl;
        }
        add((yp * 0.5) * fabs(Xp - zlog));                                  // lower triangle (pf.py:1246-1273)
        double ys = frac * (par - parp) + parp;                             // pf.py:1050
        if (!isnan_(xs)) {
            bool in_sel = ABOVE || use_all || i >= 1;
            if (y > 0.0 && in_sel) {                                        // increasing crossing
                any_inc = 1;
                if (above && !(xs <= slot[SL_LFC_X * SLOT_STRIDE])) {        // bottom LFC above the LCL (pf.py:1127-1132)
                    slot[SL_LFC_X * SLOT_STRIDE] = xs; if (!LEAN) idx()[0] = i;
                    if (!LEAN) slot[SL_LFC_T * SLOT_STRIDE] = ys;
                    slot[SL_CAPE_LFC * SLOT_STRIDE] = cape; slot[SL_CIN_LFC * SLOT_STRIDE] = cin;
                }
            }
            // top EL (pf.py:1136-1138); one at or below the LCL would be discarded by finish() anyway (pf.py:1151-1155)
            if (y < 0.0 && (ABOVE || i >= 1) && above && !(xs >= slot[SL_EL_X * SLOT_STRIDE])) {
                slot[SL_EL_X * SLOT_STRIDE] = xs; if (!LEAN) idx()[1] = i;
                if (!LEAN) slot[SL_EL_T * SLOT_STRIDE] = ys;
                slot[SL_CAPE_EL * SLOT_STRIDE] = cape;
            }
        }
        add((y * 0.5) * fabs(X - zlog));                                    // upper triangle
    }
    // LEAN: only CAPE / CIN (and the LCL / LFC / EL pressures) are wanted -- no LFC / EL temperatures, interval indices or status word:
    // they are not
    // recorded, and the lowest valid pressure is left to the caller (three LDS writes less per crossing, one per level)
    template <bool LEAN = false, bool ABOVE = false> XP_DEV void node(double P, double X, double par, double env, bool is_lcl) {
        if (is_lcl) {                                                       // the bracket is spent: SL_A* become the LFC record
            slot[SL_LFC_X * SLOT_STRIDE] = qnan(); slot[SL_LFC_T * SLOT_STRIDE] = qnan();
            slot[SL_CAPE_LFC * SLOT_STRIDE] = 0.0; slot[SL_CIN_LFC * SLOT_STRIDE] = 0.0;
        }
        // parcel minus environment is a difference of two ROUNDED temperatures in the reference: keep the compiler from
        // contracting the virtual-temperature product behind `par` into this subtraction (an fma would round once, and on
        // the knife-edge nodes of a saturated parcel -- where this difference is rounding noise of exactly the
        // reference's expressions -- that decides the sign)
        asm volatile("" : "+v"(par), "+v"(env));
        double y = par - env;
        // same sign <=> y*yp > 0 or both zero; NaN (and the first node, yp = NaN) is "not same"
        // (three comparisons and two mask operations, no branch: with the short-circuit form the compiler nested two exec-mask
        // regions here, and some lane of a wavefront has y * yp <= 0 at most levels)
        bool same = (y * yp > 0.0) | ((y == 0.0) & (yp == 0.0));
        double a = fabs(X - Xp) * ((yp + y) * 0.5);                         // pf.py:186-198
        add(same ? a : 0.0);
#ifdef XP_NODE_BALLOT
        if (__builtin_amdgcn_ballot_w64(!same) != 0ull && !same) special<LEAN, ABOVE>(X, par, env, y, a);
#else
        // (a plain divergent branch: s_and_saveexec + s_cbranch_execz skip it when no lane needs it; the explicit ballot in
        // front of it cost five instructions per node to rebuild a mask the comparison had already produced)
        if (!same) special<LEAN, ABOVE>(X, par, env, y, a);
#endif
        pos_parcel = pos_parcel || ((ABOVE || P < p_lcl) && par > env);     // pf.py:1166-1169 (ABOVE: a NaN pressure comes with a NaN parcel)
        if (!LEAN) {                                                        // status bits (LEAN: the caller did not ask for `status`)
            bad_p = bad_p || (X > Xp);                                      // NaN compares false: a missing pressure is not "bad"
            env_any = env_any || !isnan_(env);
        }
        bool pv = (ABOVE && LEAN) || !isnan_(P);                            // (ABOVE: a NaN pressure comes with a NaN parcel; the all-outputs kernels record P below)
        bool valid = pv && !isnan_(par) && !isnan_(env);                                    // p, parcel and environment all exist
        if (!LEAN && pv) slot[SL_MIN_P * SLOT_STRIDE] = P;                  // lowest valid pressure so far = the last one (LEAN: the kernel tracks the level index instead)
        top_le = (valid & (par <= env)) | (!valid & top_le);                      // (mask logic on the scalar unit: as a select the compiler round-trips the booleans through VGPRs)
        if (!LEAN) any_valid = any_valid || valid;                          // (feeds the status word only)
        if (is_lcl) { slot[SL_CAPE_LCL * SLOT_STRIDE] = cape; slot[SL_CIN_LCL * SLOT_STRIDE] = cin; }
        Xp = X; yp = y; parp = par; ++j;
    }
    struct Result { double cape, cin, lfc_p, lfc_t, el_p, el_t; int lfc_idx, el_idx, status; };
    XP_DEV Result finish(bool post_zero) {      // slot[SL_LCL_T]: the LCL (virtual) temperature, set by the caller
        Result r;
        r.status = 0;
        const double lcl_t = slot[SL_LCL_T * SLOT_STRIDE];
        int lfc_idx = idx()[0], el_idx = idx()[1];
        double lfc_p = crossing_pressure(slot[SL_LFC_X * SLOT_STRIDE]), el_p = crossing_pressure(slot[SL_EL_X * SLOT_STRIDE]);
        const double min_p = slot[SL_MIN_P * SLOT_STRIDE];
        double lfc_t = slot[SL_LFC_T * SLOT_STRIDE], el_t = slot[SL_EL_T * SLOT_STRIDE];
        // EL exists only if the parcel ends colder than the environment and the EL is above the LCL
        bool el_ok = top_le && (el_p < p_lcl);                                  // pf.py:1151-1155
        if (!el_ok) { el_p = qnan(); el_t = qnan(); el_idx = -1; }
        if (!any_valid && env_any) r.status |= 1;                               // assert of pf.py:1149
        if (bad_p) r.status |= 8;                                               // XP_ST_BAD_PRESSURE
        bool lfc_missing = any_inc == 0;
        bool replace = (pos_parcel && lfc_missing) ||
                       (!lfc_missing && isnan_(lfc_p) && (el_p < p_lcl));       // pf.py:1161-1180
        double L, cL, nL;
        if (replace) {
            L = p_lcl; cL = slot[SL_CAPE_LCL * SLOT_STRIDE]; nL = slot[SL_CIN_LCL * SLOT_STRIDE];
            lfc_p = p_lcl; lfc_t = lcl_t; lfc_idx = -2;
        } else { L = lfc_p; cL = slot[SL_CAPE_LFC * SLOT_STRIDE]; nL = slot[SL_CIN_LFC * SLOT_STRIDE]; }
        double E = el_ok ? el_p : min_p;                                        // pf.py:1329
        double cE = el_ok ? slot[SL_CAPE_EL * SLOT_STRIDE] : cape;
        r.cape = (E < L) ? RD * (cE - cL) : 0.0;                                // NaN LFC -> comparisons false -> 0.0
        r.cin = isnan_(L) ? 0.0 : RD * nL;
        if (post_zero && !(r.cin <= 0.0)) r.cin = 0.0;                          // pf.py:1387-1388
        r.lfc_p = lfc_p; r.lfc_t = lfc_t; r.el_p = el_p; r.el_t = el_t; r.lfc_idx = lfc_idx; r.el_idx = el_idx;
        return r;
    }
};

}  // namespace xp
