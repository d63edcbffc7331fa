// xp_kernels.hpp -- HIP kernels of libxparcel: one thread = one column, lanes of a wavefront own
// x-adjacent columns so every level read is one coalesced 256 B (fp32) / 512 B (fp64) request per
// array (layout (lev, y, x), col_stride == 1).  No MFMA: the path is an elementwise + per-column scan
// (SURVEY.md 8d); LDS holds the e_s / ln lookup tables and the per-thread slots of the scan (xp_device.hpp).
#pragma once
#include "xp_device.hpp"

namespace xp {

struct View { const void *data; int64_t ls, cs; };           // element strides
struct OutView { void *data; int64_t ls, cs; };

template <typename T> XP_DEV double ld(const View &v, int64_t k, int64_t c) {
    return (double)((const T *)v.data)[k * v.ls + c * v.cs];
}
template <typename T> XP_DEV T ldr(const View &v, int64_t k, int64_t c) { return ((const T *)v.data)[k * v.ls + c * v.cs]; }   // raw: no conversion at the load
template <typename T> XP_DEV double ld1(const void *p, int64_t c) { return (double)((const T *)p)[c]; }
XP_DEV void st(void *p, int f64, int64_t i, double v) {
    if (p == nullptr) return;
    if (f64) ((double *)p)[i] = v; else ((float *)p)[i] = (float)v;
}
XP_DEV void sti(int32_t *p, int64_t i, int v) { if (p) p[i] = v; }

struct ScalarsOut {
    void *cape, *cin, *lcl_p, *lcl_t, *lcl_tv, *lfc_p, *lfc_t, *el_p, *el_t;
    int32_t *lfc_idx, *el_idx, *status, *parcel_idx;
    void *par_p, *par_t, *par_td;
    int f64;
};
struct ProfileOut {
    void *v[6];            // p, t_parcel, tv_parcel, t_env, tv_env, td_env (each may be null)
    int64_t nlev_out, ls, cs;
    int f64;
    int native6;           // all six arrays wanted, in the dtype of the input views: the row is stored without per-array tests
    void *li;              // lifted index (pf.py:1722): environment minus parcel temperature of this profile at exp(li_x) hPa
    double li_x;           // ln of that pressure
};
struct CapeArgs {
    View p, t, td;
    int64_t nlev, ncol;
    const void *ex_p, *ex_t, *ex_td;      // explicit parcel
    double depth;
    int vtc, log_interp, pos_neg, post_zero, table_mode;
    int hum;                              // 1: the td view holds specific humidity (host side: picks the HUM instantiation)
    int off32;                            // 1: the three views share strides and a column's byte offset within a level fits 32 bits
    Tables tb;
    const double *es_tab;                 // e_s(T) polynomial table in global memory (staged to LDS per block)
    const double *fam_tab;                // adiabat-family table (xp::Family; family mode)
    int32_t *flags;                       // family mode: 1 = column must be redone by the RK4 kernel
    int persist;                          // family mode: persistent wavefronts (one workgroup per CU walks its share of the grid)
    int only_flagged;                     // RK4 fix-up pass: process flagged columns only
    ScalarsOut s;
    ProfileOut prof;
};

enum { PM_SURFACE = 0, PM_MU = 1, PM_ML = 2, PM_EXPLICIT = 3 };


// moisture input of one level -> dewpoint [K] (XP_HUM_SPECIFIC converts, see xparcel.h); a compile-time switch: as a
// run-time flag it cost the dewpoint path 5 VGPRs and 3 %
template <bool HUM> XP_DEV double as_dewpoint(const double *es, double p, double t, double m) {
    return HUM ? dewpoint_from_q_tab(es, p, t, m) : m;
}

struct Parcel { double p, t, td; int64_t first; int idx; bool prepend; };

// most_unstable_parcel (pf.py:102-135 with get_layer pf.py:63-100 and bound_pressure pf.py:208-227):
// highest theta-e in the lowest `depth` hPa, first maximum wins; the layer top is the level closest to
// p_bottom - depth (ties -> higher pressure).
template <typename T, bool HUM> XP_DEV Parcel select_mu_exact(const CapeArgs &a, int64_t c, const double *es, const double depth) {
    Parcel r; r.p = r.t = r.td = qnan(); r.first = a.nlev; r.idx = -1; r.prepend = false;
    double bottom = qnan(), bound = qnan(), dmin = qnan(), best = qnan();
    // one-level software prefetch: the loop is otherwise a chain of dependent HBM round trips
    // (the look-ahead values stay in the INPUT type until they are used: converting an fp32 value at the load makes the
    // wavefront wait for the load right there, and the prefetch hides nothing)
    T np_ = ldr<T>(a.p, 0, c), nt_ = ldr<T>(a.t, 0, c), ntd_ = ldr<T>(a.td, 0, c);
    for (int64_t k = 0; k < a.nlev; ++k) {
        double p = (double)np_, t = (double)nt_, td = as_dewpoint<HUM>(es, p, t, (double)ntd_);
        if (k + 1 < a.nlev) { np_ = ldr<T>(a.p, k + 1, c); nt_ = ldr<T>(a.t, k + 1, c); ntd_ = ldr<T>(a.td, k + 1, c); }
        if (isnan_(p)) continue;
        if (isnan_(bottom)) { bottom = p; bound = bottom - depth; }
        double d = fabs(p - bound);
        bool below = p < bound;
        if (below && !(d < dmin)) break;                  // the level above the bound is at least as close
        if (!(d >= dmin)) dmin = d;
        double e = ln_theta_e(es, p, t, td);               // argmax of theta_e = argmax of its logarithm
        if (!isnan_(e) && !(e <= best)) { best = e; r.p = p; r.t = t; r.td = td; r.first = k; r.idx = (int)k; }
        if (below) break;
    }
    return r;
}
// ln(theta_e) in fp32 on the hardware's log2 / exp2 / rcp: within 1.5e-6 of the fp64 value on tropospheric soundings
// (measured; tests/test_oracle_thermo.py), ~30 instructions instead of ~125.  Only used to rank levels.
XP_DEV float ln_theta_e_f32(float p, float t, float td) {
    float e = 6.112f * __builtin_amdgcn_exp2f((17.67f - 4302.645f * __builtin_amdgcn_rcpf(td - 29.65f)) * 1.4426950408889634f);
    float r = 0.6219569100577033f * e * __builtin_amdgcn_rcpf(p - e);
    float l2t = __builtin_amdgcn_logf(t), l2td = __builtin_amdgcn_logf(td);
    float tl = 56.0f + __builtin_amdgcn_rcpf(__builtin_amdgcn_rcpf(td - 56.0f) + (l2t - l2td) * (0.6931471805599453f / 800.0f));
    float l2tl = __builtin_amdgcn_logf(tl);
    return 0.6931471805599453f * (l2t + (2.0f / 7.0f) * (9.965784284662087f - __builtin_amdgcn_logf(p - e)) + 0.28f * r * (l2t - l2tl)) +
           r * (1.0f + 0.448f * r) * (3036.0f * __builtin_amdgcn_rcpf(tl) - 1.78f);
}
// The search itself: the layer's levels are ranked by the fp32 ln(theta_e); when the best level leads the runner-up by
// more than MU_F32_WINDOW (several times the fp32 error) it is the fp64 argmax too and is taken as it stands, otherwise
// (a near tie, ~0.1 % of columns) the column repeats the search in fp64 -- same parcel as the oracle either way.
constexpr float MU_F32_WINDOW = 2e-5f;
template <typename T, bool HUM> XP_DEV Parcel select_mu(const CapeArgs &a, int64_t c, const double *es, const double depth) {
    Parcel r; r.p = r.t = r.td = qnan(); r.first = a.nlev; r.idx = -1; r.prepend = false;
    double bottom = qnan(), bound = qnan(), dmin = qnan();
    float best = -__builtin_inff(), second = -__builtin_inff();
    bool any = false, odd = false;                         // odd: a level whose fp32 value is NaN / inf while the level is usable
    // The search has almost no arithmetic per level, so it runs at the speed of its loads: four levels are requested at a
    // time, the next four while these are ranked (nothing else is live in registers yet at this point of the kernel).
    constexpr int B = 4;
    T bp[B], bt[B], bd[B], cp[B], ct[B], cd[B];
    auto fetch = [&](int64_t k0, T *P_, T *T_, T *D_) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < B; ++j) {
            int64_t kk = k0 + j < a.nlev ? k0 + j : a.nlev - 1;            // clamped: rows past the top are never ranked
            P_[j] = ((const T *)a.p.data)[kk * a.p.ls + c * a.p.cs];
            T_[j] = ((const T *)a.t.data)[kk * a.t.ls + c * a.t.cs];
            D_[j] = ((const T *)a.td.data)[kk * a.td.ls + c * a.td.cs];
        }
    };
    fetch(0, bp, bt, bd);
    bool done = false;
    for (int64_t k0 = 0; k0 < a.nlev && !done; k0 += B) {
#pragma unroll
        for (int j = 0; j < B; ++j) { cp[j] = bp[j]; ct[j] = bt[j]; cd[j] = bd[j]; }
        if (k0 + B < a.nlev) fetch(k0 + B, bp, bt, bd);
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const int64_t k = k0 + j;
            if (done || k >= a.nlev) continue;
            double p = (double)cp[j], t = (double)ct[j], td = as_dewpoint<HUM>(es, p, t, (double)cd[j]);
            if (isnan_(p)) continue;
            if (isnan_(bottom)) { bottom = p; bound = bottom - depth; }
            double d = fabs(p - bound);
            bool below = p < bound;
            if (below && !(d < dmin)) { done = true; continue; }
            if (!(d >= dmin)) dmin = d;
            if (!isnan_(t) && !isnan_(td)) {
                float e = ln_theta_e_f32((float)p, (float)t, (float)td);
                if (!(e > -1e30f && e < 1e30f)) odd = true;    // out of the fp32 formula's comfort zone: decide in fp64
                if (e > best) { second = best; best = e; r.p = p; r.t = t; r.td = td; r.first = k; r.idx = (int)k; any = true; }
                else if (e > second) second = e;
            }
            if (below) done = true;
        }
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;               // the whole wavefront has left the layer
    }
    bool unsure = odd || (any && !(best - second > MU_F32_WINDOW));
    if (__builtin_amdgcn_ballot_w64(unsure) != 0ull && unsure) r = select_mu_exact<T, HUM>(a, c, es, depth);
    return r;
}

// mixed_parcel (pf.py:229-289) with mixed_layer (pf.py:137-162) / get_layer(interpolate=True):
// trapezoid in linear p of theta and w_s(p, Td) over [p_bottom - depth, p_bottom], top interpolated in ln p.
template <typename T> XP_DEV void layer_mean_step(double &sum, double p0, double v0, double p1, double v1) {
    double a = fabs(p1 - p0) * ((v0 + v1) * 0.5);
    if (!isnan_(a)) sum += a;
}
XP_DEV double interp_rule(double xb, double xa, double at, double cb, double ca) {   // pf.py:1798-1806
    double res = xb + (xa - xb) * fdiv(at - cb, ca - cb);
    return (xb == xa) ? xb : res;
}
template <typename T, bool HUM> XP_DEV Parcel select_ml(const CapeArgs &a, int64_t c, const double *es, const double depth) {
    Parcel r; r.p = r.t = r.td = qnan(); r.first = a.nlev; r.idx = -1; r.prepend = true;
    double p_start = ld<T>(a.p, 0, c);
    double bottom = qnan(), top = qnan();
    double s_th = 0.0, s_w = 0.0;
    double pp = qnan(), thp = qnan(), wp = qnan();        // previous row of the layer
    double pb = qnan(), thb = qnan(), wb = qnan();        // last row with a valid pressure >= top
    bool closed = false;
    T np_ = ldr<T>(a.p, 0, c), nt_ = ldr<T>(a.t, 0, c), ntd_ = ldr<T>(a.td, 0, c);   // one-level software prefetch, in the input type (see select_mu_exact)
    for (int64_t k = 0; k < a.nlev; ++k) {
        double p = (double)np_, t = (double)nt_, td = as_dewpoint<HUM>(es, p, t, (double)ntd_);
        if (k + 1 < a.nlev) { np_ = ldr<T>(a.p, k + 1, c); nt_ = ldr<T>(a.t, k + 1, c); ntd_ = ldr<T>(a.td, k + 1, c); }
        if (isnan_(bottom) && !isnan_(p)) { bottom = p; top = bottom - depth; }
        if (!isnan_(p) && p < top) {
            // insert the interpolated top row, close the integral; the profile continues from this level
            double lt = flog(top), cb = flog(pb), ca = flog(p);
            double th_a = t / fpow(p / 1000.0, KAPPA), w_a = sat_mix(p, td);
            double cb2 = cb, ca2 = ca, tha = th_a, wa = w_a;
            if (pb == top) { ca2 = cb; tha = thb; wa = wb; }
            double th_t = interp_rule(thb, tha, lt, cb2, ca2), w_t = interp_rule(wb, wa, lt, cb2, ca2);
            layer_mean_step<T>(s_th, pp, thp, top, th_t);
            layer_mean_step<T>(s_w, pp, wp, top, w_t);
            r.first = k; closed = true;
            break;
        }
        double th = t / fpow(p / 1000.0, KAPPA), w = sat_mix(p, td);
        if (k > 0) { layer_mean_step<T>(s_th, pp, thp, p, th); layer_mean_step<T>(s_w, pp, wp, p, w); }
        pp = p; thp = th; wp = w;
        if (!isnan_(p)) { pb = p; thb = th; wb = w; }
    }
    if (!closed) {
        // column never gets above the layer top: the inserted row holds NaN unless a level sits exactly on it
        double th_t = (pb == top) ? thb : qnan(), w_t = (pb == top) ? wb : qnan();
        layer_mean_step<T>(s_th, pp, thp, top, th_t);
        layer_mean_step<T>(s_w, pp, wp, top, w_t);
    }
    double dlay = fabs(top - bottom);
    double th_m = (1.0 / dlay) * s_th, w_m = (1.0 / dlay) * s_w;
    r.p = p_start;                                                         // pf.py:250, 287
    r.t = th_m * fpow(p_start / 1000.0, KAPPA);                             // pf.py:268-269
    r.td = dewpoint_of_e(vapor_pressure(p_start, w_m));                    // pf.py:275-280
    return r;
}

// ---------------------------------------------------------------------------------------------
// cape_cin (pf.py:1394-1475) and its drivers, fused: parcel selection, LCL, parcel profile with
// the LCL as a virtual level, LFC/EL, CAPE/CIN, optional profile output.
//
// The level loop runs in two wave-uniform phases.  Phase A (some lane of the wavefront is still at or below
// its LCL; decided with a ballot) carries the full logic: dry or moist parcel, bracketing levels for the
// environment at the LCL, the LCL node -- one node per lane and iteration, see `source` below.  Phase B (every lane
// above its LCL) is the steady state and only advances the moist adiabat, so the LCL machinery costs nothing for most
// of the column.
// MODE: 0 = exact by RK4, 1 = reference lookup tables, 2 = exact by the adiabat family (columns it cannot serve are
// flagged and redone by a MODE 0 launch with only_flagged set).
// HUM: the moisture view holds specific humidity (XP_HUM_SPECIFIC).
// DEF: the reference's default option set -- virtual-temperature correction on, LCL environment interpolated in ln p, sign-filtered sums (pf.py:1396, 1293) --
// as compile-time constants: the selects and scalar registers the run-time switches cost in the level loop go away.
// Instantiated for the dewpoint-input kernels of modes 0 / 1; family mode and every other combination take DEF = false
// (xp_cape_tu.hip has the dispatch rule and why).
// LEAN (with DEF): the caller wants neither LFC / EL temperatures nor interval indices (the bench, the gather of a
// multi-GPU run): they are not tracked, see Scan::node.
// PROFILE + DEF + LEAN (family mode only): the lifted index of the profile and none of its rows -- `LAZY` below.
template <typename T, int PMODE, bool PROFILE, int MODE, bool HUM, bool DEF, bool LEAN, bool PERSIST>
__global__ __launch_bounds__(XP_CAPE_THREADS, (XP_CAPE_THREADS >= 1024 ? 4 : MODE == 2 ? 3 : PROFILE ? 4 : (PMODE == PM_SURFACE ? (HUM ? 3 : 1) : 4))) void k_cape_cin(CapeArgs a) {
    // Occupancy: four wavefronts per SIMD everywhere -- the family translation units through their 1024-thread workgroups
    // (one per CU), the others through the bound (256-thread workgroups, four per CU).  The bound only steers the
    // allocator (the dewpoint-input surface kernel of modes 0 / 1 lands below 128 VGPRs on its own and is allocated worse
    // when forced); tests/test_kernel_resources.py checks what comes out.
    constexpr bool TABLE = (MODE == 1), FAMILY = (MODE == 2);
    // LAZY (family mode, PROFILE + DEF + LEAN: the host picks it when the caller wants the lifted index, NO profile rows and
    // nothing beyond CAPE / CIN -- the product bundle's three parcel passes): the parcel's plain temperature -- three Newton
    // steps from its virtual temperature at every level above the LCL, a quarter of a profile kernel's instructions -- is
    // found only at the two nodes that bracket the lifted-index level.
    constexpr bool LAZY = FAMILY && PROFILE && LEAN;
    // One LDS object with the e_s / ln table FIRST: at LDS address 0 the table base folds into the immediate offsets of the
    // ds_read instructions (the table is read ~15 times per level; behind the slots every access paid a v_mov for the base).
    // (Not for the family kernels with profile output: they sit at the 128-VGPR cap, and with the one-object layout the
    // register allocator spilled 18 instead of 12 VGPRs there -- c3 in family mode 27.2 -> 30.8 ms.  They keep separate arrays.)
    struct Lds { double es[LDS_TAB]; double fam[FAMILY ? FAM_SIZE : 1]; double slot[SLOT_FIELDS * SLOT_STRIDE]; int next; };
    struct LdsRef { double *es, *fam, *slot; int *next; } lds;
    if constexpr (FAMILY && PROFILE) {
        __shared__ double l_es[LDS_TAB];
        __shared__ double l_fam[FAM_SIZE];
        __shared__ double l_slot[SLOT_FIELDS * SLOT_STRIDE];
        __shared__ int l_next;
        lds.es = l_es; lds.fam = l_fam; lds.slot = l_slot; lds.next = &l_next;
    } else {
        __shared__ Lds l;
        lds.es = l.es; lds.fam = l.fam; lds.slot = l.slot; lds.next = &l.next;
    }
    double *const s_es = lds.es;
    // family mode: the coefficient table lives in LDS too (46.7 KB; read 81 doubles at a time by lanes that differ only in
    // their psi-piece: broadcast + adjacent banks, conflict-free, ~100 cycles instead of an L2 round trip per batch)
    double *const s_fam = lds.fam;
    if (FAMILY) for (int i = threadIdx.x; i < FAM_SIZE; i += blockDim.x) s_fam[i] = a.fam_tab[i];
    int &s_next = *lds.next;                                               // PERSIST: the workgroup's next tile
    if (PERSIST && threadIdx.x == 0) s_next = (int)(blockDim.x >> 6);
    // PERSIST (family mode, large grids; the host decides): the grid is one workgroup per CU, the tables are staged once,
    // and every wavefront takes 64-column tiles from an atomic counter until the grid is done -- no staging and no drain
    // between workgroups, and whichever wavefront is free takes the next tile, so the chip walks the grid roughly in
    // order (8-Mi-column configs: -3 % surface, -12 % mixed-layer, -15 % most-unstable; c2 has only four tiles per
    // wavefront and is 4 % faster as an ordinary launch).
    constexpr bool persist = PERSIST;
    int64_t c0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.only_flagged) {                                                  // fix-up pass: most blocks have nothing to do
        int need = (c0 < a.ncol) ? a.flags[c0] : 0;
        if (!__syncthreads_or(need)) return;
        const double *es0 = stage_es_table(a.es_tab, s_es);
        (void)es0;
        if (!need) return;
    } else {
        stage_es_table(a.es_tab, s_es);
        if (!persist && c0 >= a.ncol) return;
    }
    const double *es = s_es;
    double *const s_slot = lds.slot;

    auto column = [&](const int64_t c) __attribute__((always_inline)) {
    Parcel pc;
    if (PMODE == PM_SURFACE) {
        pc.p = ld<T>(a.p, 0, c); pc.t = ld<T>(a.t, 0, c); pc.td = as_dewpoint<HUM>(es, pc.p, pc.t, ld<T>(a.td, 0, c));
        pc.first = 0; pc.idx = 0; pc.prepend = false;
    } else if (PMODE == PM_EXPLICIT) {
        pc.p = ld1<T>(a.ex_p, c); pc.t = ld1<T>(a.ex_t, c); pc.td = ld1<T>(a.ex_td, c);
        pc.first = 0; pc.idx = -1; pc.prepend = false;
    } else if (PMODE == PM_MU) {
        pc = select_mu<T, HUM>(a, c, es, a.depth);
    } else {
        pc = select_ml<T, HUM>(a, c, es, a.depth);
    }

    const bool vtc = DEF || (a.vtc != 0), pos_neg = DEF || (a.pos_neg != 0), log_interp = DEF || (a.log_interp != 0);
    const bool need_w = vtc || PROFILE;
    const Lcl l = lcl(pc.p, pc.t, pc.td);
    int status = l.not_converged ? 2 : 0;
    const ScalarsOut &s = a.s;

    if (isnan_(l.p)) {
        // NaN parcel / LCL blanks the whole profile (pf.py:965-985): CAPE = CIN = 0.0, everything else NaN
        if (PROFILE) {
            for (int64_t j = 0; j < a.prof.nlev_out; ++j)
                for (int v = 0; v < 6; ++v) st(a.prof.v[v], a.prof.f64, j * a.prof.ls + c * a.prof.cs, qnan());
            st(a.prof.li, a.prof.f64, c, qnan());
        }
        st(s.cape, s.f64, c, 0.0); st(s.cin, s.f64, c, 0.0);
        st(s.lcl_p, s.f64, c, l.p); st(s.lcl_t, s.f64, c, l.t); st(s.lcl_tv, s.f64, c, l.tv);
        st(s.lfc_p, s.f64, c, qnan()); st(s.lfc_t, s.f64, c, qnan()); st(s.el_p, s.f64, c, qnan()); st(s.el_t, s.f64, c, qnan());
        sti(s.lfc_idx, c, -1); sti(s.el_idx, c, -1); sti(s.status, c, status); sti(s.parcel_idx, c, pc.idx);
        st(s.par_p, s.f64, c, pc.p); st(s.par_t, s.f64, c, pc.t); st(s.par_td, s.f64, c, pc.td);
        if (FAMILY) a.flags[c] = 0;
        return;
    }

    // everything known before the scan is stored now, so that it does not occupy registers through the level loop
    st(s.lcl_p, s.f64, c, l.p); st(s.lcl_t, s.f64, c, l.t); st(s.lcl_tv, s.f64, c, l.tv);
    sti(s.parcel_idx, c, pc.idx);
    st(s.par_p, s.f64, c, pc.p); st(s.par_t, s.f64, c, pc.t); st(s.par_td, s.f64, c, pc.td);
    const double vf_parcel = need_w ? virt_factor_tab(es, pc.t, pc.td, pc.p, false) : 1.0;   // 1 + 0.608 w of the parcel (pf.py:748, 767)
    // ln p bookkeeping.  Levels use the table logarithm; the LCL node uses the library log (its crossing tests
    // "p* < p_lcl" then break ties as on the CPU); a level that sits exactly on the LCL pressure takes the LCL's
    // value so that the interval between the two stays zero-width; and the parcel's own ln p (x0) is whatever its
    // level gets, so that the surface parcel reproduces its level bit for bit (T0 * exp(0)) -- the reference's lfc_el
    // branches on that exact equality (pf.py:1117-1120).
    const double x_lcl = log(l.p);
    const double x0 = (pc.p == l.p) ? x_lcl : log_tab<true>(es, pc.p);

    Scan sc; sc.init(l.p, x_lcl, pos_neg, s_slot + threadIdx.x);
    sc.slot[SL_LCL_T * SLOT_STRIDE] = vtc ? l.tv : l.t;                  // pf.py:1442 / 1461
    Moist m;
    Family fam;
    if (FAMILY) fam.start(s_fam, es, l.p, x_lcl, l.t, l.tv);
    else m.start(es, l.p, x_lcl, l.t, TABLE, a.tb);
    double fam_off = l.tv - l.t;                                            // family profile kernels: Tv - T of the parcel at the node before (temperature_from)
    double fam_offp = fam_off;                                              // ... and at the node before that (temperature_from2)

    int jout = 0;                                                           // profile row
    double li_d = qnan();                                                   // PROFILE: environment minus parcel temperature of the node before this one (lifted index)
    // LAZY: the node before this one as it came -- its parcel temperature where the node had one (dry adiabat, LCL), else
    // its virtual temperature, to be inverted if the next node turns out to close the bracket
    double lz_te = qnan(), lz_tq = qnan(), lz_p = qnan();
    int lz_known = 1;
    bool li_done = false;
    // CAPE / CIN-only kernels and the lowest valid pressure of the profile (stands in for a missing EL, pf.py:1329): recovered
    // after the walk (below) -- except for the mixed-layer parcel, whose kernels the register allocator serves better with
    // the per-node bookkeeping (17-27 spilled VGPRs otherwise): the level index of the last valid-pressure node
    constexpr bool TRACK = LEAN && PMODE == PM_ML;
    int last_k = -1, cur_k = -1;
    // `above` (a std::integral_constant): this node and the one before it lie strictly above the LCL (phase B)
    auto emit = [&](auto above, double P, double X, double tp, double tvp, double te, double tve, double tde, bool is_lcl) __attribute__((always_inline)) {
        if (PROFILE) {
            if (!LAZY && jout < a.prof.nlev_out) {
                int64_t o = jout * a.prof.ls + c * a.prof.cs;
                bool dead = isnan_(P);                                     // NaN-coordinate rows come out all-NaN (pf.py:963, 988)
#ifndef XP_NO_NATIVE6
                if (a.prof.native6) {
                    // the common request (the drivers, BASELINE config 3): six stores of the input dtype, no null / dtype
                    // test per array (each was two scalar branches plus, at this register pressure, two v_readlane), the
                    // NaN-row select done on the converted value
                    const T vP = (T)P;
                    ((T *)a.prof.v[0])[o] = vP;
                    ((T *)a.prof.v[1])[o] = dead ? vP : (T)tp;
                    ((T *)a.prof.v[2])[o] = dead ? vP : (T)tvp;
                    ((T *)a.prof.v[3])[o] = dead ? vP : (T)te;
                    ((T *)a.prof.v[4])[o] = dead ? vP : (T)tve;
                    ((T *)a.prof.v[5])[o] = dead ? vP : (T)tde;
                } else
#endif
                {
                st(a.prof.v[0], a.prof.f64, o, P);
                st(a.prof.v[1], a.prof.f64, o, dead ? P : tp);
                st(a.prof.v[2], a.prof.f64, o, dead ? P : tvp);
                st(a.prof.v[3], a.prof.f64, o, dead ? P : te);
                st(a.prof.v[4], a.prof.f64, o, dead ? P : tve);
                st(a.prof.v[5], a.prof.f64, o, dead ? P : tde);
                }
            }
            ++jout;
            // lifted_index (pf.py:1722 = log_interp of the profile's two temperatures at one pressure, pf.py:1813): the
            // nodes come with decreasing pressure, so the first one at or above the level closes the bracket that the
            // node before it opened (coords_before / coords_after of pf.py:1774-1775; a NaN-pressure row is no
            // coordinate; value rule of pf.py:1802-1806)
            // -- both temperatures take the same weight, so their difference is interpolated: one value of state.
            // The state lives in an LDS slot where the workgroup has one to spare (not the 1024-thread family build).
            if constexpr (LAZY) {
                // (a node without a parcel temperature of its own hands in NaN for it; a NaN parcel inverts to NaN)
                const bool known = !isnan_(tp);
                if (!li_done && X <= a.prof.li_x + 1e-12) {
                    const bool on = X >= a.prof.li_x - 1e-12;
                    double off0 = 0.0, off1 = 0.0;
                    const double tc = known ? tp : Family::temperature_from(es, P, tvp, off0);
                    const double tb4 = lz_known ? lz_tq : Family::temperature_from(es, lz_p, lz_tq, off1);
                    const double d_ = te - tc, dp = lz_te - tb4;
                    const double wgt = (a.prof.li_x - sc.Xp) / (X - sc.Xp);
                    st(a.prof.li, a.prof.f64, c, (on || dp == d_) ? d_ : dp + (d_ - dp) * wgt);
                    li_done = true;
                }
                if (!isnan_(P)) { lz_te = te; lz_tq = known ? tp : tvp; lz_p = P; lz_known = known ? 1 : 0; }
            } else if (a.prof.li) {
                constexpr bool LI_SLOT = SLOT_FIELDS > SL_LI;
                const double d_ = te - tp;
                if (!li_done && X <= a.prof.li_x + 1e-12) {
                    // a node ON the level (the table logarithm and the host's differ in the last bits) is its own bracket
                    const bool on = X >= a.prof.li_x - 1e-12;
                    const double dp = LI_SLOT ? sc.slot[(LI_SLOT ? SL_LI : 0) * SLOT_STRIDE] : li_d;
                    const double wgt = (a.prof.li_x - sc.Xp) / (X - sc.Xp);
                    st(a.prof.li, a.prof.f64, c, (on || dp == d_) ? d_ : dp + (d_ - dp) * wgt);
                    li_done = true;
                }
                if (!isnan_(P)) { if (LI_SLOT) sc.slot[(LI_SLOT ? SL_LI : 0) * SLOT_STRIDE] = d_; else li_d = d_; }
            }
        }
        if (TRACK) {
            if (!isnan_(P)) { last_k = (is_lcl || cur_k < 0) ? -1 : cur_k; if (is_lcl || cur_k < 0) sc.slot[SL_MIN_P * SLOT_STRIDE] = P; }
        }
        sc.template node<LEAN, decltype(above)::value>(P, X, vtc ? tvp : tp, vtc ? tve : te, is_lcl);
    };

    bool lcl_done = false;
    // last valid-pressure node at or below the LCL (the lower bracket of the LCL interpolation): LDS slots, written in
    // phase A only
    double *const br = sc.slot;
    br[SL_BR_P * SLOT_STRIDE] = qnan(); br[SL_BR_X * SLOT_STRIDE] = qnan(); br[SL_BR_T * SLOT_STRIDE] = qnan(); br[SL_BR_TD * SLOT_STRIDE] = qnan();
    // environment at the LCL: bracketing-level interpolation in ln p or p (pf.py:897-906, 1758-1811) between the last
    // valid level at or below the LCL (the br slots) and the first level above it
    auto lcl_environment = [&](double pa, double xa, double ta, double tda, double &te, double &tde) __attribute__((always_inline)) {
        double at = log_interp ? x_lcl : l.p;
        const double pb = br[SL_BR_P * SLOT_STRIDE], xb = br[SL_BR_X * SLOT_STRIDE], tb_ = br[SL_BR_T * SLOT_STRIDE], tdb = br[SL_BR_TD * SLOT_STRIDE];
        lds_wait_all();
        double cb = log_interp ? xb : pb, ca = log_interp ? xa : pa;
        double ta2 = ta, tda2 = tda;
        if (pb == l.p) { ca = cb; ta2 = tb_; tda2 = tdb; }                 // a level sits exactly on the LCL
        te = interp_rule(tb_, ta2, at, cb, ca); tde = interp_rule(tdb, tda2, at, cb, ca);
    };
    // parcel temperature / mixing ratio above the LCL; e_s(T) rides along with the RK4 state in exact mode
    // `Q`: with specific-humidity input and no profile output the environment's mixing ratio is q / (1 - q) itself --
    // the reference's w = RH * w_s(p, T) with RH = e_s(Td) / e_s(T) (pf.py:698-704) undoes exactly the q -> Td chain --
    // so above the LCL neither the dewpoint nor the two e_s evaluations are needed (m_ then holds q, not Td)
    auto moist_node = [&](double P, double X, double T_, double m_, bool Q) __attribute__((always_inline)) {
        // family mode: the table holds the parcel's VIRTUAL temperature; the plain temperature is derived from it only
        // where somebody wants it (profile output, no virtual-temperature correction)
        // (family CAPE/CIN-only kernels: the Horner chain of the parcel's virtual temperature is evaluated together with the two
        // e_s chains of the environment, below: three independent chains interleaved)
        // (surface / explicit parcels: c2 -2.2 % same-box; the searching parcels' kernels, at 127-128 VGPRs, gain nothing)
        constexpr bool TRIO = FAMILY && LEAN && !PROFILE && !HUM && (PMODE == PM_SURFACE || PMODE == PM_EXPLICIT);
        bool f_top = false, f_any_top = false;
        const double fz = TRIO ? fam.locate(X, f_top, f_any_top) : 0.0;
        double tf = TRIO ? 0.0 : FAMILY ? fam.at(X) : m.at(P, X, a.tb, true);     // NaN pressure -> NaN
        // one wave-uniform range test per level instead of one per e_s evaluation; the two sides are separate code (the
        // asm barrier keeps the compiler from merging them into one path full of selects)
        constexpr bool PARCEL_ES = TABLE;                                  // exact mode: e_s(T) rides along with the RK4 state
        // (family profile kernels: the parcel's plain temperature is found by Newton steps that evaluate e_s a few kelvin
        // below its virtual temperature -- same promise, with that margin)
        constexpr bool FAM_T = FAMILY && PROFILE;
        unsigned dist = table_dist(T_);
        if (!Q) dist = umax_(dist, table_dist(m_));
        if (PARCEL_ES && need_w) dist = umax_(dist, table_dist(tf));
        const bool in_range = all_in_table(dist) && (!FAM_T || LAZY || in_table(tf, 8.0));
        double ep = 0.0, tve, tpf = 0.0;
        if (__builtin_amdgcn_ballot_w64(!in_range) == 0ull) {
            if (need_w && !FAMILY) ep = PARCEL_ES ? es_tab(es, tf, true) : m.e;
            if constexpr (TRIO) {                                          // (LEAN implies the default options: need_w)
                double e_td, e_t;
                es_tab2_horner(es, m_, T_, fam.c, fz, e_td, e_t, tf);
                tve = T_ * __builtin_fma(e_td * frcp1(P - e_t), VT_EPS * EPS, 1.0);        // = virt_env_tab, bit for bit
            } else {
            tve = !need_w ? T_ : Q ? virt(T_, (m_ > 0.0 && m_ < 1.0) ? fdiv(m_, 1.0 - m_) : qnan()) : virt_env_tab<LEAN && !PROFILE>(es, T_, m_, P, true);
            }
            if (FAM_T) tpf = LAZY ? qnan() : Family::temperature_from2(es, P, tf, fam_off, fam_offp, true);
        } else {
            if constexpr (TRIO) tf = fam.horner(fz);
            double tq = tf;
            asm volatile("" : "+v"(tq));
            if (need_w && !FAMILY) ep = PARCEL_ES ? es_tab(es, tq, false) : m.e;
            tve = !need_w ? T_ : Q ? virt(T_, (m_ > 0.0 && m_ < 1.0) ? fdiv(m_, 1.0 - m_) : qnan()) : virt_env_tab(es, T_, m_, P, false);
            if (FAM_T) tpf = LAZY ? qnan() : Family::temperature_from2(es, P, tq, fam_off, fam_offp, false);
        }
        if constexpr (TRIO) { if (f_any_top) tf = fam.top_value(tf, X, f_top); }
        double tp, tvp;
        if (FAMILY) {
            tvp = tf;
            tp = PROFILE ? tpf : !vtc ? Family::temperature_of(es, P, tf) : tf;   // (not used when neither holds)
        } else {
            tp = tf;
            tvp = need_w ? virt(tp, mix_of_e(ep, P)) : tp;                                // pf.py:760
        }
        emit(std::true_type{}, P, X, tp, tvp, T_, tve, m_, false);
    };
    // Phase A: the wavefront's columns sit on both sides of their LCLs.  Every lane feeds exactly ONE node per iteration:
    //   below its LCL       the level that was just loaded (dry adiabat);
    //   crossing            the LCL node instead of that level (environment interpolated, pf.py:897-920) -- the level
    //                       waits in (sP, sT, sM) and the lane is one level behind the loads from here on ("skewed");
    //   above (skewed)      the level that has been waiting, while the one just loaded takes its place.
    // So the node evaluation (environment e_s twice, the scan) runs once per iteration whatever the lanes are doing;
    // inserting the LCL node as a second node of the same iteration made it run twice in almost every iteration of phase A
    // (some lane of 64 crosses at nearly every level there: 13.4 against 6.2 us per level and Mi-column, measured).
    // One iteration past the top level (`last`, nothing loaded) flushes the waiting level; a column whose LCL lies above
    // the top level feeds its LCL node there, with no upper bracket (NaN environment).
    double sP = qnan(), sT = qnan(), sM = qnan();
    // specific-humidity input, mixed-layer parcel: the prepended parcel node carries a DEWPOINT; when it is the waiting
    // level (a supersaturated mixed parcel lies above its own LCL) it must not be converted again, and it has to be fed by
    // phase A, not by phase B's q-based shortcut
    constexpr bool PREP = HUM && PMODE == PM_ML;
    bool s_is_td = false;
    auto source = [&](auto raw, double Pc, double Tc, double Mc, bool last, int kc) __attribute__((always_inline)) {
        const bool skew = lcl_done;
        double P = skew ? sP : Pc, T_ = skew ? sT : Tc;
        const double M_ = skew ? sM : Mc;
        double Td_ = (decltype(raw)::value && !(PREP && skew && s_is_td)) ? as_dewpoint<HUM>(es, P, T_, M_) : M_;
        if (fabs(P - l.p) <= LCL_SNAP * l.p) P = l.p;                       // on the LCL (see xp::lcl)
        double X = log_tab<true>(es, P);
        X = (P == l.p) ? x_lcl : X;
        if (TRACK) cur_k = skew ? kc - 1 : kc;
        const bool cross = !skew && (last || P < l.p);
        if (!LEAN && isnan_(P) && !skew && !last) status |= 4;             // NaN pressure below the LCL (see xparcel.h)
        // only the parcel temperature / mixing ratio is branched, the environment and the scan node are shared
        double tp, tvp;
        if (!skew) {                                                       // dry adiabat (pf.py:313, 767)
            tp = pc.t * dry_factor(es, KAPPA * (X - x0));
            tvp = need_w ? tp * vf_parcel : tp;
        } else if (FAMILY) {                                               // the table holds the virtual temperature
            tvp = fam.at(X);
            if (PROFILE && !LAZY) fam_offp = fam_off;
            tp = LAZY ? qnan() : PROFILE ? Family::temperature_from(es, P, tvp, fam_off) : !vtc ? Family::temperature_of(es, P, tvp) : tvp;
        } else {
            tp = m.at(P, X, a.tb);
            tvp = need_w ? virt(tp, mix_of_e(TABLE ? es_tab(es, tp) : m.e, P)) : tp;
        }
        if (cross) {                                                       // (a plain divergent branch: saveexec + execz)          // this lane's node is its LCL
            double te, tde;
            lcl_environment(P, X, T_, Td_, te, tde);
            // without profile output the scan only sees the temperature picked by the correction switch, which sits in
            // the SL_LCL_T slot: neither LCL temperature has to stay in registers through the loop
            const double lsel = br[SL_LCL_T * SLOT_STRIDE];
            P = l.p; X = x_lcl; T_ = te; Td_ = tde;
            tp = PROFILE ? l.t : lsel; tvp = PROFILE ? l.tv : lsel;
        }
        double tve = T_;                                                   // pf.py:839-843, 911-920
        if (need_w) {                                                      // one wave-uniform range test for the two e_s, as in phase B
            if (__builtin_amdgcn_ballot_w64(!all_in_table(umax_(table_dist(T_), table_dist(Td_)))) == 0ull) tve = virt_env_tab<LEAN && !PROFILE>(es, T_, Td_, P, true);
            else { double tq = T_; asm volatile("" : "+v"(tq)); tve = virt_env_tab(es, tq, Td_, P, false); }
        }
        // For a saturated parcel (LCL == parcel level) the sign of parcel-minus-environment at the LCL node is rounding
        // noise of exactly the reference's expressions: those columns evaluate them in its operation order.
        const bool tie = need_w && cross && (l.p == pc.p);
        if (tie) { double q = T_; asm volatile("" : "+v"(q)); tve = virt_ref(q, Td_, l.p); }
        // A level exactly ON the LCL pairs the dry temperature with the saturation mixing ratio at the moist-adiabat
        // temperature (pf.py:773 uses <=).  For a saturated parcel this is the parcel's own level and the same holds.
        const bool on_lcl = need_w && !cross && (P == l.p);
        if (on_lcl) {
            double ta = FAMILY ? l.t : m.at(P, X, a.tb);
            asm volatile("" : "+v"(ta));
            double ea = es_ref(ta);
            tvp = tp * (1.0 + VT_EPS * (EPS * ea / (P - ea)));
            tve = virt_ref(T_, Td_, P);
        }
        emit(std::false_type{}, P, X, tp, tvp, T_, tve, Td_, cross);
        if (!isnan_(P) && !skew && !cross) { br[SL_BR_P * SLOT_STRIDE] = P; br[SL_BR_X * SLOT_STRIDE] = X; br[SL_BR_T * SLOT_STRIDE] = T_; br[SL_BR_TD * SLOT_STRIDE] = Td_; }
        lcl_done = skew || cross;
        sP = Pc; sT = Tc; sM = Mc;
        if (PREP) s_is_td = !decltype(raw)::value;
    };

    if (pc.prepend) source(std::false_type{}, pc.p, pc.t, pc.td, false, -1);  // ML: the parcel is the new level 0 (pf.py:1641-1644)
    constexpr bool SEARCH = PMODE == PM_MU || PMODE == PM_ML;
    int k = (int)pc.first;      // per lane for MU / ML parcels through phase A; phase B re-aligns the wavefront (below)
    // software-prefetched level loop
    // Three per-lane row pointers that WALK up the levels: set once (64-bit multiply-add, a quarter-rate instruction),
    // then advanced by the row stride with two full-rate adds per array and level.
    const int64_t lane_off = (int64_t)c * a.p.cs * (int64_t)sizeof(T);
    const int64_t row_step = a.p.ls * (int64_t)sizeof(T);
    // (address space 1 = global, spelled out: behind the asm barrier below the compiler would otherwise fall back to
    // flat loads, which also count against the LDS counter and so make every LDS wait a memory wait)
    typedef const char __attribute__((address_space(1))) *GPtr;
    GPtr lp = nullptr, lt = nullptr, ld_ = nullptr;
    // (the look-ahead buffer holds the values as they are in memory: an fp32 level is converted when it is TAKEN -- converting
    // at the load made every fp32 kernel wait for its loads on the spot, i.e. run without any prefetch)
    T np_ = (T)qnan(), nt_ = (T)qnan(), ntd_ = (T)qnan();
    auto seek = [&](int64_t kk) __attribute__((always_inline)) {             // the next load3() reads level kk
        const int64_t o = kk * row_step + lane_off;
        lp = (GPtr)a.p.data + o; lt = (GPtr)a.t.data + o; ld_ = (GPtr)a.td.data + o;
    };
    auto load3 = [&](T &P_, T &T2_, T &Td2_) __attribute__((always_inline)) {
        typedef const T __attribute__((address_space(1))) *GT;
        P_ = *(GT)lp; T2_ = *(GT)lt; Td2_ = *(GT)ld_;
        lp += row_step; lt += row_step; ld_ += row_step;
        asm volatile("" : "+v"(lp), "+v"(lt), "+v"(ld_));                  // (keeps the walk: no re-derivation from the level index)
    };
    const int nlev = (int)a.nlev;                                          // the host checks nlev < 2^31: 32-bit scalar compares in the loops
    seek(k);
    if (k < nlev) load3(np_, nt_, ntd_);
    // Level out of the look-ahead buffer, the next one requested.  The buffer itself holds NaN once the levels are used up
    // (no select per level), and the wave-uniform loops count DOWN -- `rem` levels not yet taken out of the buffer -- so
    // that they compare against 0 and 1 and the level count is not live in them (it used to be spilled and read back with
    // eight v_readlane per level).
    auto take = [&](bool more, double &P_, double &T2_, double &M_) __attribute__((always_inline)) {
        // (one wait for the three values: left alone the compiler waits for each one just before its copy)
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0x0F70); __builtin_amdgcn_sched_barrier(0);
        M_ = (double)ntd_; T2_ = (double)nt_; P_ = (double)np_;            // (the value requested last first)
        if (more) load3(np_, nt_, ntd_);
        else { np_ = (T)qnan(); nt_ = (T)qnan(); ntd_ = (T)qnan(); }
    };
    constexpr bool Q = HUM && !PROFILE;
    for (; k <= nlev; ++k) {                                             // phase A (the searching parcels: per-lane level index)
        if (__ballot(!lcl_done || (PREP && s_is_td)) == 0ull) break;       // wave-uniform: everybody is above its LCL
        double P, T_, M_;
        take(k + 1 < nlev, P, T_, M_);
        source(std::true_type{}, P, T_, M_, k >= nlev, k);
    }
    // (counting phase A down as well costs the surface / explicit-parcel kernels 60-70 spilled VGPRs at the 128 cap: measured)
    // Phase B: every lane is past its LCL and one level behind the loads: the level in (sP, sT, sM) is fed while the next
    // one arrives; the iteration past the top level feeds the last one.
    // The plain walk: the waiting stage of phase A (sP, sT, sM: level k - 1) is no longer needed -- every lane is one level
    // behind the loads -- and carrying it cost three register moves per level.  The waiting level goes back into the
    // look-ahead buffer (the request for level k that is in flight is dropped and made again: one row per tile, from L2),
    // and from then on a level goes from the buffer straight into the node.  Needs (sP, sT, sM) to be a LEVEL (exact in T).
    auto plain_walk = [&](int k_) __attribute__((always_inline)) {
        int rem = __builtin_amdgcn_readfirstlane(nlev - k_);               // levels not yet requested into the buffer ...
        if (rem > 0) { lp -= row_step; lt -= row_step; ld_ -= row_step; }
        // (the dropped request is waited for HERE, once: left pending on phase A's buffer registers it made the compiler put
        // `s_waitcnt vmcnt` in front of the first reuse of those registers inside the loop -- in the middle of every iteration,
        // where it waited for the loads of THIS level's successor: half the prefetch distance gone)
#ifndef XP_NO_PREWAIT
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0x0F70); __builtin_amdgcn_sched_barrier(0);
#endif
        np_ = (T)sP; nt_ = (T)sT; ntd_ = (T)sM;
        rem += 1;                                                          // ... nodes still to feed: levels k - 1 ... nlev - 1
        asm volatile("" : "+s"(rem));
        for (; rem > 0; --rem, ++k_) {
            double P, T_, M_;
            // (take() without its NaN refill: what the buffer holds after the last level is never looked at here; the value
            // requested last is copied first, so that the compiler's one wait covers all three)
            M_ = (double)ntd_; T_ = (double)nt_; P = (double)np_;
            if (rem > 1) load3(np_, nt_, ntd_);
            if (TRACK) cur_k = k_ - 1;
            moist_node(P, log_tab<true>(es, P), T_, Q ? M_ : as_dewpoint<HUM>(es, P, T_, M_), Q);
        }
    };
    if constexpr (SEARCH) {
        // ... with a WAVE-UNIFORM level index.  The columns of a searching parcel start at their own levels, so after
        // phase A the lanes stand on different levels and every load would touch as many level rows as there are distinct
        // positions (the most-unstable kernel fetched 3.7 x its algorithmic bytes: 6.4 L2 requests per load instead of 2).
        // From here the wavefront walks up from its lowest lane and a lane sits out until the walk reaches its own level;
        // the lowest lane decides the number of iterations either way.
        const int resume = k;                                              // the level this lane would load next
        int ku = nlev + 1;
        for (int probe = 0; probe <= nlev; ++probe) if (__ballot(resume <= probe) != 0ull) { ku = probe; break; }
        seek(ku);
        np_ = (T)qnan(); nt_ = (T)qnan(); ntd_ = (T)qnan();
        if (ku < nlev) load3(np_, nt_, ntd_);
        int rem = nlev - ku;
        asm volatile("" : "+s"(rem));
        // (Leaving this gated loop for the plain walk once every lane has joined was measured: mixed-layer +- 0, most-unstable
        // + 3.7 %, and the profile-output kernels spill over a hundred VGPRs with two copies of the node code.)
        for (; rem >= 0; --rem, ++ku) {
            double Pn, Tn, Mn;
            take(rem > 1, Pn, Tn, Mn);
            if (ku >= resume) {
                if (TRACK) cur_k = ku - 1;
                moist_node(sP, log_tab<true>(es, sP), sT, Q ? sM : as_dewpoint<HUM>(es, sP, sT, sM), Q);
                sP = Pn; sT = Tn; sM = Mn;
            }
        }
    } else {
        plain_walk(k);
    }
    if (TRACK) {
        if (last_k >= 0) sc.slot[SL_MIN_P * SLOT_STRIDE] = ld<T>(a.p, last_k, c);
    } else if (LEAN) {
        // The lowest valid pressure of the profile (what stands in for a missing EL, pf.py:1329) is not tracked per node in
        // the CAPE / CIN-only kernels: pressures decrease along the profile, so it is the smaller of the LCL pressure and the
        // last level with a valid pressure -- found now by looking down from the top (one load per lane unless the top is
        // missing).  (The mixed-layer parcel's own node lies at or below its LCL, so it never is the minimum.)
        double pm = l.p;
        bool found = false;
        if constexpr (!SEARCH) {
            // after the plain walk the look-ahead buffer still holds the top level (the last one taken; nothing was requested
            // behind it): when its pressure is valid -- the normal case -- that is the answer, without a load whose round
            // trip nothing would cover at this point
            const double q = (double)np_;
            if (!isnan_(q)) { pm = (q < pm) ? q : pm; found = true; }
        }
        for (int kk = nlev - 1; kk >= (int)pc.first; --kk) {
            if (__ballot(!found) == 0ull) break;
            const double q = ld<T>(a.p, kk, c);
            if (!found && !isnan_(q)) { pm = (q < pm) ? q : pm; found = true; }
        }
        sc.slot[SL_MIN_P * SLOT_STRIDE] = pm;
    }
    if (PROFILE) {
        for (; jout < a.prof.nlev_out; ++jout) {
            int64_t o = jout * a.prof.ls + c * a.prof.cs;
            for (int v = 0; v < 6; ++v) st(a.prof.v[v], a.prof.f64, o, qnan());
        }
        if (!li_done) st(a.prof.li, a.prof.f64, c, qnan());              // the profile never reaches the level
    }

    // The output pointers are fetched from the kernel arguments only now, through a pointer the compiler cannot see
    // through, so that it does not load all of them up front and carry ~26 scalar registers across the level loop
    // (where they were being spilled into VGPR lanes and read back lane by lane: 840 v_readlane in the family kernel).
    // `a` is the kernel's only parameter, so it sits at offset 0 of the kernarg segment; taking &a instead would make the
    // compiler copy the whole struct to scratch.
    typedef const CapeArgs __attribute__((address_space(4))) *KernargPtr;
    KernargPtr late = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(late) : : "memory");
    const int of64 = late->s.f64;
    Scan::Result r = sc.finish(late->post_zero != 0);
    status |= r.status;
    if (FAMILY) late->flags[c] = fam.bad ? 1 : 0;
    st(late->s.cape, of64, c, r.cape); st(late->s.cin, of64, c, r.cin);
    st(late->s.lfc_p, of64, c, r.lfc_p); st(late->s.lfc_t, of64, c, r.lfc_t);
    st(late->s.el_p, of64, c, r.el_p); st(late->s.el_t, of64, c, r.el_t);
    sti(late->s.lfc_idx, c, r.lfc_idx); sti(late->s.el_idx, c, r.el_idx); sti(late->s.status, c, status);
    };   // column

    if (PERSIST) {
        // Every workgroup owns a contiguous share of the grid's 64-column tiles and its wavefronts take them one by one
        // from a counter in LDS.  (One device-wide counter in memory was measured first: same-address atomics execute at
        // the memory side one after the other, ~50 ns each -- 12 000 of them are most of c2's 0.7 ms.)
        const int64_t ntiles = (a.ncol + 63) >> 6;
        const int t0 = (int)(ntiles * blockIdx.x / gridDim.x), t1 = (int)(ntiles * (blockIdx.x + 1) / gridDim.x);
        int tile = t0 + (int)(threadIdx.x >> 6);                            // s_next starts behind these (set before the staging barrier)
        while (tile < t1) {
            const int64_t c = ((int64_t)tile << 6) + (threadIdx.x & 63);
            if (c < a.ncol) column(c);
            if ((threadIdx.x & 63) == 0) tile = t0 + atomicAdd(&s_next, 1);
            tile = __builtin_amdgcn_readfirstlane(tile);
        }
    } else {
        column(c0);
    }
}

// parcels only (most_unstable_parcel pf.py:102, mixed_parcel pf.py:229)
template <typename T, int PMODE> __global__ __launch_bounds__(256) void k_select_parcel(CapeArgs a) {
    __shared__ double s_es[LDS_TAB];
    const double *es = stage_es_table(a.es_tab, s_es);
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncol) return;
    Parcel pc = (PMODE == PM_MU) ? select_mu<T, false>(a, c, es, a.depth) : select_ml<T, false>(a, c, es, a.depth);
    st(a.s.par_p, a.s.f64, c, pc.p); st(a.s.par_t, a.s.f64, c, pc.t); st(a.s.par_td, a.s.f64, c, pc.td);
    sti(a.s.parcel_idx, c, pc.idx);
}

// mixed_layer (pf.py:137-162) of one variable
template <typename T> __global__ __launch_bounds__(256)
void k_mixed_layer(View pv, View vv, int64_t nlev, int64_t ncol, double depth_in, void *out, int f64) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double bottom = qnan(), top = qnan(), s = 0.0, pp = qnan(), vp = qnan(), pb = qnan(), vb = qnan();
    bool closed = false;
    for (int64_t k = 0; k < nlev; ++k) {
        double p = ld<T>(pv, k, c), v = ld<T>(vv, k, c);
        if (isnan_(bottom) && !isnan_(p)) { bottom = p; top = bottom - depth_in; }
        if (!isnan_(p) && p < top) {
            double cb = flog(pb), ca = flog(p), va = v;
            if (pb == top) { ca = cb; va = vb; }
            layer_mean_step<T>(s, pp, vp, top, interp_rule(vb, va, flog(top), cb, ca));
            closed = true;
            break;
        }
        if (k > 0) layer_mean_step<T>(s, pp, vp, p, v);
        pp = p; vp = v;
        if (!isnan_(p)) { pb = p; vb = v; }
    }
    if (!closed) layer_mean_step<T>(s, pp, vp, top, (pb == top) ? vb : qnan());
    st(out, f64, c, (1.0 / fabs(top - bottom)) * s);
}

// lcl (pf.py:609-682) on n parcels
template <typename T> __global__ __launch_bounds__(256)
void k_lcl(int64_t n, const void *p, const void *t, const void *td, void *op, void *ot, void *otv, int32_t *status) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    Lcl l = lcl(ld1<T>(p, c), ld1<T>(t, c), ld1<T>(td, c));
    const int f64 = sizeof(T) == 8;
    st(op, f64, c, l.p); st(ot, f64, c, l.t); st(otv, f64, c, l.tv); sti(status, c, l.not_converged ? 2 : 0);
}

// dry_lapse (pf.py:291-316); parcel pressure defaults to the column maximum
template <typename T> __global__ __launch_bounds__(256)
void k_dry_lapse(View pv, int64_t nlev, int64_t ncol, const void *pt, const void *pp, OutView out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double t0 = ld1<T>(pt, c), p0 = qnan();
    if (pp) p0 = ld1<T>(pp, c);
    else for (int64_t k = 0; k < nlev; ++k) { double p = ld<T>(pv, k, c); if (!isnan_(p) && !(p <= p0)) p0 = p; }
    for (int64_t k = 0; k < nlev; ++k) {
        double p = ld<T>(pv, k, c);
        st(out.data, sizeof(T) == 8, k * out.ls + c * out.cs, t0 * fpow(p / p0, KAPPA));
    }
}

// moist_lapse (pf.py:525-607): levels in any order relative to the parcel pressure.  Levels at or above the
// reference (p <= p_ref) are marched upwards in level order, levels below it downwards in reverse level order
// (pressure must decrease with level index on each side, the reference's input contract).
template <typename T> __global__ __launch_bounds__(256)
void k_moist_lapse(View pv, int64_t nlev, int64_t ncol, const void *pt, const void *pp, int table_mode, Tables tb,
                   const double *es_g, OutView out) {
    __shared__ double s_es[LDS_TAB];
    const double *es = stage_es_table(es_g, s_es);
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    double t0 = ld1<T>(pt, c);
    double p0 = pp ? ld1<T>(pp, c) : ld<T>(pv, 0, c);                      // pf.py:549-550
    double x0 = flog(p0);
    Moist m; m.start(es, p0, x0, t0, table_mode != 0, tb);
    for (int64_t k = 0; k < nlev; ++k) {
        double p = ld<T>(pv, k, c);
        if (p <= p0) st(out.data, f64, k * out.ls + c * out.cs, m.at(p, flog(p), tb));
        else if (isnan_(p)) st(out.data, f64, k * out.ls + c * out.cs, qnan());
    }
    m.start(es, p0, x0, t0, table_mode != 0, tb);
    for (int64_t k = nlev - 1; k >= 0; --k) {
        double p = ld<T>(pv, k, c);
        if (p > p0) st(out.data, f64, k * out.ls + c * out.cs, m.at(p, flog(p), tb));
    }
}

// parcel_profile (pf.py:712-780) without the LCL level
template <typename T> __global__ __launch_bounds__(256)
void k_parcel_profile(View pv, int64_t nlev, int64_t ncol, const void *pp, const void *pt, const void *ptd,
                      int table_mode, Tables tb, const double *es_g, OutView ot, OutView otv, void *olp, void *olt,
                      void *oltv) {
    __shared__ double s_es[LDS_TAB];
    const double *es = stage_es_table(es_g, s_es);
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    double p0 = ld1<T>(pp, c), t0 = ld1<T>(pt, c), td0 = ld1<T>(ptd, c);
    Lcl l = lcl(p0, t0, td0);
    double w_parcel = mixing_ratio(t0, td0, p0);
    double x_lcl = flog(l.p);
    Moist m; m.start(es, l.p, x_lcl, l.t, table_mode != 0, tb);
    for (int64_t k = 0; k < nlev; ++k) {
        double P = ld<T>(pv, k, c);
        double tp, w;
        if (P >= l.p) {
            tp = t0 * fpow(P / p0, KAPPA);
            w = (P == l.p) ? mix_of_e(sat_vapor_pressure(l.t), P) : w_parcel;
        } else {
            tp = m.at(P, flog(P), tb);
            w = mix_of_e(sat_vapor_pressure(tp), P);
        }
        if (ot.data) st(ot.data, f64, k * ot.ls + c * ot.cs, tp);
        if (otv.data) st(otv.data, f64, k * otv.ls + c * otv.cs, virt(tp, w));
    }
    st(olp, f64, c, l.p); st(olt, f64, c, l.t); st(oltv, f64, c, l.tv);
}

// lfc_el (pf.py:1066-1198) on caller-supplied profiles: the same state machine, fed directly
template <typename T> __global__ __launch_bounds__(XP_CAPE_THREADS)
void k_lfc_el(View pv, View parv, View envv, int64_t nlev, int64_t ncol, const void *lcl_p, const void *lcl_t,
              ScalarsOut s) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double lp = ld1<T>(lcl_p, c), lt = ld1<T>(lcl_t, c);
    __shared__ double s_slot[SLOT_FIELDS * SLOT_STRIDE];
    Scan sc; sc.init(lp, log(lp), true, s_slot + threadIdx.x);
    sc.slot[SL_LCL_T * SLOT_STRIDE] = lt;
    for (int64_t k = 0; k < nlev; ++k) {
        double P = ld<T>(pv, k, c);
        sc.node(P, flog(P), ld<T>(parv, k, c), ld<T>(envv, k, c), false);
    }
    Scan::Result r = sc.finish(false);
    st(s.lfc_p, s.f64, c, r.lfc_p); st(s.lfc_t, s.f64, c, r.lfc_t); st(s.el_p, s.f64, c, r.el_p); st(s.el_t, s.f64, c, r.el_t);
    sti(s.lfc_idx, c, r.lfc_idx); sti(s.el_idx, c, r.el_idx); sti(s.status, c, r.status);
}

// cape_cin_base (pf.py:1291-1392) on caller-supplied profiles and LFC/EL: direct form (LFC and EL are
// known up front, so every area is tested against them as the reference does, no snapshots).
template <typename T> __global__ __launch_bounds__(256)
void k_cape_cin_base(View pv, View envv, View parv, int64_t nlev, int64_t ncol, const void *lfc_p, const void *el_p,
                     int pos_neg, int post_zero, void *cape, void *cin) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    const int f64 = sizeof(T) == 8;
    double L = ld1<T>(lfc_p, c), E = ld1<T>(el_p, c);
    if (isnan_(E)) {                                                       // pf.py:1329
        for (int64_t k = 0; k < nlev; ++k) { double p = ld<T>(pv, k, c); if (!isnan_(p) && !(p >= E)) E = p; }
    }
    double sc_ = 0.0, sn = 0.0, Pp = qnan(), Xp = qnan(), yp = qnan();
    for (int64_t k = 0; k < nlev; ++k) {
        double P = ld<T>(pv, k, c), X = flog(P), y = ld<T>(parv, k, c) - ld<T>(envv, k, c);
        if (k > 0) {
            bool ynan = isnan_(y) || isnan_(yp);
            double s0 = (double)((yp > 0.0) - (yp < 0.0)), s1 = (double)((y > 0.0) - (y < 0.0));
            bool handled = false;
            if (ynan || s1 != s0) {
                double xs = (y * Xp - yp * X) / (y - yp);
                double zy = ((xs - Xp) / (X - Xp)) * (y - yp) + yp;
                if (!isnan_(zy)) {
                    handled = true;
                    double zlog = flog(fexp(xs));
                    double dx = Xp - zlog, a0 = (yp * 0.5) * fabs(dx), pm = fexp(Xp - dx * 0.5);
                    if (pm <= L && pm >= E && (!pos_neg || a0 > 0.0)) sc_ += a0;
                    if (pm >= L && (!pos_neg || a0 < 0.0)) sn += a0;
                    dx = X - zlog; a0 = (y * 0.5) * fabs(dx); pm = fexp(X - dx * 0.5);
                    if (pm <= L && pm >= E && (!pos_neg || a0 > 0.0)) sc_ += a0;
                    if (pm >= L && (!pos_neg || a0 < 0.0)) sn += a0;
                }
            }
            if (!handled) {
                double a0 = fabs(X - Xp) * ((yp + y) * 0.5);
                if (!isnan_(a0)) {
                    if (Pp <= L && Pp >= E && P <= L && P >= E && (!pos_neg || a0 > 0.0)) sc_ += a0;
                    if (Pp >= L && P >= L && (!pos_neg || a0 < 0.0)) sn += a0;
                }
            }
        }
        Pp = P; Xp = X; yp = y;
    }
    double cin_v = RD * sn;
    if (post_zero && !(cin_v <= 0.0)) cin_v = 0.0;
    st(cape, f64, c, RD * sc_); st(cin, f64, c, cin_v);
}

// wet_bulb_temperature (pf.py:389-445): one thread per (level, column) element
template <typename T> __global__ __launch_bounds__(256)
void k_wet_bulb(View pv, View tv, View tdv, int64_t nlev, int64_t ncol, int table_mode, Tables tb, const double *es_g,
                OutView out) {
    __shared__ double s_es[LDS_TAB];
    const double *es = stage_es_table(es_g, s_es);
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nlev * ncol) return;
    int64_t k = e / ncol, c = e - k * ncol;
    double p = ld<T>(pv, k, c), t = ld<T>(tv, k, c), td = ld<T>(tdv, k, c);
    Lcl l = lcl(p, t, td);                                                 // pf.py:420-422
    Moist m; m.start(es, l.p, log(l.p), l.t, table_mode != 0, tb);
    double r = qnan();
    if (!isnan_(p)) {
        if (p == l.p) r = m.at(p, m.x, tb);                                // saturated: LCL snapped onto the element
        else r = m.at(p, log(p), tb);                                      // pf.py:425-428 (moist descent, p > p_lcl)
    }
    st(out.data, sizeof(T) == 8, k * out.ls + c * out.cs, r);
}

// linear_interp / log_interp (pf.py:1758-1828) of one variable at one coordinate per column; no ordering assumed
template <typename T> __global__ __launch_bounds__(256)
void k_interp_level(View cv, View xv, int64_t nlev, int64_t ncol, const void *at_p, int at_scalar, int log_coords,
                    void *out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double at = ld1<T>(at_p, at_scalar ? 0 : c);
    if (log_coords) at = clog(at);
    // coords_before = smallest coordinate >= at, coords_after = largest <= at (pf.py:1774-1775); values = mean over
    // the levels that carry exactly that coordinate, skipping NaN (pf.py:1798-1799)
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
    double xb = nb ? sb / (double)nb : qnan(), xa = na ? sa / (double)na : qnan();
    double res = xb + (xa - xb) * ((at - cb) / (ca - cb));
    st(out, sizeof(T) == 8, c, (xb == xa) ? xb : res);                     // pf.py:1802-1806
}

// The same rule for NV variables at NT coordinates in ONE pass over the column (the product bundle interpolates
// temperature, dewpoint and height to 850 / 700 / 500 hPa: seven launches of k_interp_level that each re-read the
// pressure).  Per (variable, coordinate) exactly k_interp_level's arithmetic; out[v * NT + j], null = not wanted.
struct InterpMany { View x[4]; double at[4]; void *out[16]; };
template <typename T, int NV, int NT> __global__ __launch_bounds__(256)
void k_interp_levels(View cv, InterpMany m, int64_t nlev, int64_t ncol, int log_coords) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double at[NT], cb[NT], ca[NT], sb[NV][NT], sa[NV][NT];
    int nb[NV][NT], na[NV][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        at[j] = log_coords ? clog(m.at[j]) : m.at[j];
        cb[j] = ca[j] = qnan();
#pragma unroll
        for (int v = 0; v < NV; ++v) { sb[v][j] = sa[v][j] = 0.0; nb[v][j] = na[v][j] = 0; }
    }
    for (int64_t k = 0; k < nlev; ++k) {
        double cc = ld<T>(cv, k, c), x[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) x[v] = ld<T>(m.x[v], k, c);
        if (log_coords) cc = clog(cc);
        if (isnan_(cc)) continue;
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
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            double xb = nb[v][j] ? sb[v][j] / (double)nb[v][j] : qnan(), xa = na[v][j] ? sa[v][j] / (double)na[v][j] : qnan();
            double res = xb + (xa - xb) * ((at[j] - cb[j]) / (ca[j] - cb[j]));
            st(m.out[v * NT + j], sizeof(T) == 8, c, (xb == xa) ? xb : res);
        }
}

// dewpoint_from_specific_humidity (MetPy 1.4.1, parcel_test.py:262-266): one thread per element
template <typename T> __global__ __launch_bounds__(256)
void k_dewpoint_from_q(View pv, View tv, View qv, int64_t nlev, int64_t ncol, OutView out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nlev * ncol) return;
    int64_t k = e / ncol, c = e - k * ncol;
    st(out.data, sizeof(T) == 8, k * out.ls + c * out.cs, dewpoint_from_q(ld<T>(pv, k, c), ld<T>(tv, k, c), ld<T>(qv, k, c)));
}

// freezing_level_height (pf.py:2137-2158): the smallest x among ALL intersections (find_intersections pf.py:992-1064,
// linear x) of a(x) with the constant `value`, NaN when there is none.  An interval counts when sign(a - value)
// changes or is NaN at either end (pf.py:1019-1022); its intersection is NaN unless all four numbers are finite.
template <typename T> __global__ __launch_bounds__(256)
void k_crossing_level(View xv, View av, int64_t nlev, int64_t ncol, double value, void *out) {
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncol) return;
    double best = qnan(), x0 = qnan(), d0 = qnan();
    for (int64_t k = 0; k < nlev; ++k) {
        double x1 = ld<T>(xv, k, c), d1 = ld<T>(av, k, c) - value;
        if (k > 0) {
            double s0 = (double)((d0 > 0.0) - (d0 < 0.0)), s1 = (double)((d1 > 0.0) - (d1 < 0.0));
            if (isnan_(d0) || isnan_(d1) || s0 != s1) {
                double xi = (d1 * x0 - d0 * x1) / (d1 - d0);               // pf.py:1046
                if (!isnan_(xi) && !(xi >= best)) best = xi;
            }
        }
        x0 = x1; d0 = d1;
    }
    st(out, sizeof(T) == 8, c, best);
}

// mixing_ratio (pf.py:684-710): RH = e_s(Td) / e_s(T) times the saturation mixing ratio at (p, T) -- MetPy 1.4.1's
// relative_humidity_from_dewpoint + mixing_ratio_from_relative_humidity, in the reference's operation order
template <typename T> __global__ __launch_bounds__(256)
void k_mixing_ratio(View tv, View tdv, View pv, int64_t nlev, int64_t ncol, OutView out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nlev * ncol) return;
    int64_t k = e / ncol, c = e - k * ncol;
    double t = ld<T>(tv, k, c), td = ld<T>(tdv, k, c), p = ld<T>(pv, k, c);
    double est = es_ref(t);
    st(out.data, sizeof(T) == 8, k * out.ls + c * out.cs, (es_ref(td) / est) * (EPS * est / (p - est)));
}

// dense (nlev, ncol) copy of a strided view: the fall-back for inputs whose three views do not share strides (or
// whose column offsets exceed 32 bits), see CapeArgs::off32
template <typename T> __global__ __launch_bounds__(256)
void k_densify(View v, int64_t nlev, int64_t ncol, T *out) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nlev * ncol) return;
    int64_t k = e / ncol, c = e - k * ncol;
    out[e] = ((const T *)v.data)[k * v.ls + c * v.cs];
}

// host-side launcher of k_cape_cin<T, pm, profile, MODE, a.hum>, defined in xp_cape_tu.hip -- one translation unit per
// (T, MODE), so that the 96 instantiations of the big kernel compile in parallel
template <typename T, int MODE> void launch_cape_mode(const CapeArgs &a, int pm, bool profile, hipStream_t s);

}  // namespace xp
