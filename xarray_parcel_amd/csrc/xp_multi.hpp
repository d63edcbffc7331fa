// xp_multi.hpp -- several parcels of ONE grid lifted in one pass (adiabat-family moist mode).
//
// The reference's products lift two or three parcels from the same three arrays -- most-unstable and mixed-layer CAPE / CIN
// of BASELINE config 5 (pf.py:1557, 1651), most-unstable + 100 hPa + 50 hPa mixed-layer in conv_properties (pf.py:1984-2006) --
// and every single-parcel launch of k_cape_cin re-reads p / T / Td and recomputes what does not depend on the parcel: ln p
// and the environment's virtual temperature, the two e_s evaluations of pf.py:839-843 -- about 60 of the ~150 fp64
// instructions a level costs.  Here one thread still owns one column, but carries NP parcel "chains" (LCL, dry-adiabat
// constants, xp::Family coefficients, xp::Scan state, LDS slots) through ONE walk over the levels:
//
//   * first every chain, one after the other, does what depends on its parcel alone: the parcel search, the LCL, the
//     adiabat's label, and the levels up to the LCL with the LCL node in their midst (k_cape_cin's phase A, on per-lane
//     level indices: ~10-15 iterations until the whole wavefront is past its LCLs);
//   * then ONE walk up the remaining levels (k_cape_cin's phase B, ~85 % of a column): every level is loaded once, with a
//     wave-uniform index (one coalesced row request per array), its environment node (ln p, Tv) is evaluated once, and each
//     chain that has reached it feeds it to its scan with its own moist-adiabat value.  A chain that resumes higher up sits
//     out until the walk reaches its level.
// (A first version walked all chains through their LCLs in the shared loop as well: with the most-unstable parcels of
// 18 % of the columns starting at levels 12-43 some lane of nearly every wavefront was below its LCL up to level ~50, the
// expensive below-LCL node ran there for everybody, and the fused c5 step took 39 ms against 24.3 for two separate calls.)
//
// Every chain performs exactly the floating-point operations of the single-parcel kernel on its nodes (same device
// functions, same order), so the results are bit-identical to separate xp_cape_cin calls (tests/test_gpu_multi.py).
// Columns a chain's family table cannot serve are flagged per chain and redone by the single-parcel RK4 kernel.
// Workgroups: XP_CAPE_THREADS threads, one per CU (LDS: e_s / ln tables 11.8 KB + family table 46.7 KB + NP x
// SLOT_FIELDS x XP_CAPE_THREADS slot doubles); two parcels at 512 threads = 156.9 KB, two wavefronts per SIMD with up
// to 256 VGPRs each -- the chains of a thread are independent dependency chains, which is where the latency hiding
// that the lower occupancy gives up comes back from.
#pragma once
#include "xp_kernels.hpp"

namespace xp {

constexpr int MULTI_MAX = 3;
struct MultiArgs {
    CapeArgs base;                    // views, shape, options, tables, persist (base.s / base.flags / base.depth / base.prof unused)
    int np;
    int mode[MULTI_MAX];              // PM_SURFACE | PM_MU | PM_ML
    double depth[MULTI_MAX];
    ScalarsOut s[MULTI_MAX];
    int32_t *flags[MULTI_MAX];        // 1 = the column of this chain must be redone by the RK4 kernel
    void *li[MULTI_MAX];              // lifted index per chain (pf.py:1722), nullable
    double li_x;                      // ln of its pressure
    int li_f64;
};

struct Lev { double P, X, T, Td, tve; };      // one level with its environment node: pressure, ln p, T, Td, Tv (or T)

struct Chain {
    double lp, xl, lt;     // LCL pressure, its logarithm, LCL temperature
    double pt, x0, vfac;   // dry adiabat below the LCL: parcel temperature, ln of its pressure, 1 + 0.608 x its mixing ratio (pf.py:748)
    int first;             // first level of the grid that belongs to this chain's profile (INT_MAX: blank chain)
    int status;
    bool done;             // the LCL node has been fed: the chain consumes `prev` from now on
    bool sat;              // LCL on the parcel's own level (saturated parcel)
    double li_p, li_e, li_q;   // lifted index: last valid-pressure node (pressure, environment T, parcel T or -Tv where only Tv is known)
    bool li_done;
    Scan sc;
    Family fam;
};

template <typename T, int NP, bool PERSIST>
__global__ __launch_bounds__(XP_CAPE_THREADS) void k_cape_cin_multi(MultiArgs a) {
    static_assert(NP >= 1 && NP <= MULTI_MAX, "1..3 parcels");
    __shared__ double s_es[LDS_TAB];
    __shared__ double s_fam[FAM_SIZE];
    for (int i = threadIdx.x; i < FAM_SIZE; i += blockDim.x) s_fam[i] = a.base.fam_tab[i];
    __shared__ int s_next;
    if (PERSIST && threadIdx.x == 0) s_next = (int)(blockDim.x >> 6);
    stage_es_table(a.base.es_tab, s_es);
    const int64_t c0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!PERSIST && c0 >= a.base.ncol) return;
    const double *es = s_es;
    __shared__ double s_slot[NP * SLOT_FIELDS * SLOT_STRIDE];
    constexpr int DEAD = 0x7fffffff;

    auto column = [&](const int64_t c) __attribute__((always_inline)) {
    const CapeArgs &b = a.base;
    const bool vtc = b.vtc != 0, pos_neg = b.pos_neg != 0;
    const bool need_w = vtc;
    const int nlev = (int)b.nlev;
    Chain h0, h1, h2;
    auto each = [&](auto f) __attribute__((always_inline)) {
        f(h0, std::integral_constant<int, 0>{});
        if constexpr (NP > 1) f(h1, std::integral_constant<int, 1>{});
        if constexpr (NP > 2) f(h2, std::integral_constant<int, 2>{});
    };
    auto env_tv = [&](double T_, double Td_, double P) __attribute__((always_inline)) {          // pf.py:839-843 (rare paths: own range test)
        return need_w ? virt_env_tab(es, T_, Td_, P, false) : T_;
    };

    // One node of chain h below / at / just above its LCL -- the logic of k_cape_cin's phase A (`source`), on one level
    // (P, T_, Td_): `skew` = the chain is past its LCL and this is the level that has been waiting; otherwise it is the level
    // just loaded, and when it lies above the LCL (or nothing is left: `last`) the LCL node is fed in its place.
    auto feed = [&](Chain &h, double P, double T_, double Td_, const bool skew, const bool last) __attribute__((always_inline)) {
        double *const br = h.sc.slot;
        if (fabs(P - h.lp) <= LCL_SNAP * h.lp) P = h.lp;                           // on the LCL (see xp::lcl)
        double X = log_tab<true>(es, P);
        X = (P == h.lp) ? h.xl : X;
        const bool cross = !skew && (last || P < h.lp);
        if (isnan_(P) && !skew && !last) h.status |= 4;                            // NaN pressure below the LCL (see xparcel.h)
        double tp, tvp;
        if (!skew) {                                                               // dry adiabat (pf.py:313, 767)
            tp = h.pt * dry_factor(es, KAPPA * (X - h.x0));
            tvp = need_w ? tp * h.vfac : tp;
        } else {                                                                   // the table holds the virtual temperature
            tvp = h.fam.at(X);
            tp = !vtc ? Family::temperature_of(es, P, tvp) : tvp;
        }
        if (__builtin_amdgcn_ballot_w64(cross) != 0ull && cross) {                  // this lane's node is its LCL
            // environment at the LCL: bracketing-level interpolation in ln p or p (pf.py:897-906, 1758-1811)
            const double at = b.log_interp ? h.xl : h.lp;
            const double pb = br[SL_BR_P * SLOT_STRIDE], xb = br[SL_BR_X * SLOT_STRIDE], tb_ = br[SL_BR_T * SLOT_STRIDE], tdb = br[SL_BR_TD * SLOT_STRIDE];
            double cb = b.log_interp ? xb : pb, ca = b.log_interp ? X : P;
            double ta2 = T_, tda2 = Td_;
            if (pb == h.lp) { ca = cb; ta2 = tb_; tda2 = tdb; }                    // a level sits exactly on the LCL
            const double te = interp_rule(tb_, ta2, at, cb, ca), tde = interp_rule(tdb, tda2, at, cb, ca);
            const double lsel = br[SL_LCL_T * SLOT_STRIDE];
            P = h.lp; X = h.xl; T_ = te; Td_ = tde;
            tp = lsel; tvp = lsel;
        }
        double tve = T_;                                                           // pf.py:839-843, 911-920
        if (need_w) {                                                              // one wave-uniform range test for the two e_s
            if (__builtin_amdgcn_ballot_w64(!(in_table(T_, 0.0) && in_table(Td_, 0.0))) == 0ull) tve = virt_env_tab(es, T_, Td_, P, true);
            else { double tq = T_; asm volatile("" : "+v"(tq)); tve = virt_env_tab(es, tq, Td_, P, false); }
        }
        const bool tie = need_w && cross && h.sat;
        if (__builtin_amdgcn_ballot_w64(tie) != 0ull && tie) { double q = T_; asm volatile("" : "+v"(q)); tve = virt_ref(q, Td_, h.lp); }
        const bool on_lcl = need_w && !cross && (P == h.lp);                       // pf.py:773 uses <=
        if (__builtin_amdgcn_ballot_w64(on_lcl) != 0ull && on_lcl) {
            double ta = h.lt;
            asm volatile("" : "+v"(ta));
            double ea = es_ref(ta);
            tvp = tp * (1.0 + VT_EPS * (EPS * ea / (P - ea)));
            tve = virt_ref(T_, Td_, P);
        }
        h.sc.template node<false, false>(P, X, vtc ? tvp : tp, vtc ? tve : T_, cross);
        if (!isnan_(P) && !skew && !cross) { br[SL_BR_P * SLOT_STRIDE] = P; br[SL_BR_X * SLOT_STRIDE] = X; br[SL_BR_T * SLOT_STRIDE] = T_; br[SL_BR_TD * SLOT_STRIDE] = Td_; }
        h.done = skew || cross;
    };

    const int64_t lane_off = (int64_t)c * b.p.cs * (int64_t)sizeof(T), row_step = b.p.ls * (int64_t)sizeof(T);
    typedef const char __attribute__((address_space(1))) *GPtr;
    typedef const T __attribute__((address_space(1))) *GT;

    // ---- per chain, one after the other: parcel, LCL, label, and the levels up to the LCL ---------------------------------
    // (k_cape_cin's phase A: every lane walks from ITS first level with its own row pointers until the whole wavefront is
    // past its LCLs -- ~10-15 iterations; these rows are read again by the shared walk below, out of L2)
    each([&](Chain &h, auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const int mode = a.mode[i];
        Parcel pc;
        if (mode == PM_SURFACE) {
            pc.p = ld<T>(b.p, 0, c); pc.t = ld<T>(b.t, 0, c); pc.td = ld<T>(b.td, 0, c);
            pc.first = 0; pc.idx = 0; pc.prepend = false;
        } else if (mode == PM_MU) {
            pc = select_mu<T, false>(b, c, es, a.depth[i]);
        } else {
            pc = select_ml<T, false>(b, c, es, a.depth[i]);
        }
        const Lcl l = lcl(pc.p, pc.t, pc.td);
        const ScalarsOut &s = a.s[i];
        double *const slot = s_slot + i * (SLOT_FIELDS * SLOT_STRIDE) + threadIdx.x;
        h.status = l.not_converged ? 2 : 0;
        h.lp = l.p; h.lt = l.t; h.xl = qnan(); h.pt = pc.t; h.x0 = qnan(); h.vfac = 1.0;
        h.sat = false; h.done = true; h.first = DEAD;
        h.li_p = h.li_e = h.li_q = qnan(); h.li_done = false;
        h.fam.tab = s_fam; h.fam.q = 0; h.fam.s = 0.0; h.fam.bad = false; h.fam.poison();
        if (isnan_(l.p)) {
            // NaN parcel / LCL blanks the whole profile (pf.py:965-985): CAPE = CIN = 0.0, everything else NaN
            h.sc.init(l.p, qnan(), pos_neg, slot);
            st(s.cape, s.f64, c, 0.0); st(s.cin, s.f64, c, 0.0);
            st(s.lcl_p, s.f64, c, l.p); st(s.lcl_t, s.f64, c, l.t); st(s.lcl_tv, s.f64, c, l.tv);
            st(s.lfc_p, s.f64, c, qnan()); st(s.lfc_t, s.f64, c, qnan()); st(s.el_p, s.f64, c, qnan()); st(s.el_t, s.f64, c, qnan());
            sti(s.lfc_idx, c, -1); sti(s.el_idx, c, -1); sti(s.status, c, h.status); sti(s.parcel_idx, c, pc.idx);
            st(s.par_p, s.f64, c, pc.p); st(s.par_t, s.f64, c, pc.t); st(s.par_td, s.f64, c, pc.td);
            st(a.li[i], a.li_f64, c, qnan());
            a.flags[i][c] = 0;
        } else {
            st(s.lcl_p, s.f64, c, l.p); st(s.lcl_t, s.f64, c, l.t); st(s.lcl_tv, s.f64, c, l.tv);
            sti(s.parcel_idx, c, pc.idx);
            st(s.par_p, s.f64, c, pc.p); st(s.par_t, s.f64, c, pc.t); st(s.par_td, s.f64, c, pc.td);
            h.vfac = need_w ? virt_factor_tab(es, pc.t, pc.td, pc.p, false) : 1.0; // pf.py:748
            // ln p bookkeeping as in k_cape_cin: library log for the LCL, table logarithm for levels, and the parcel's own
            // ln p is whatever its level gets (the surface parcel reproduces its level bit for bit, pf.py:1117-1120)
            h.xl = log(l.p);
            h.x0 = (pc.p == l.p) ? h.xl : log_tab<true>(es, pc.p);
            h.sat = (l.p == pc.p);
            h.sc.init(l.p, h.xl, pos_neg, slot);
            slot[SL_LCL_T * SLOT_STRIDE] = vtc ? l.tv : l.t;                       // pf.py:1442 / 1461
            h.fam.start(s_fam, es, l.p, h.xl, l.t, l.tv);
            slot[SL_BR_P * SLOT_STRIDE] = qnan(); slot[SL_BR_X * SLOT_STRIDE] = qnan(); slot[SL_BR_T * SLOT_STRIDE] = qnan(); slot[SL_BR_TD * SLOT_STRIDE] = qnan();
            h.done = false;
            h.first = (int)pc.first;
        }
        // mixed layer: the parcel is the new level 0 of its profile (pf.py:1641-1644)
        const bool pre = pc.prepend && h.first != DEAD;
        if (__builtin_amdgcn_ballot_w64(pre) != 0ull && pre) {
            feed(h, pc.p, pc.t, pc.td, false, false);
            // a supersaturated mixed parcel lies above its own LCL: the LCL node went first and the parcel node follows it
            const bool again = h.done;
            if (__builtin_amdgcn_ballot_w64(again) != 0ull && again) feed(h, pc.p, pc.t, pc.td, true, false);
        }
        // levels first, first + 1, ... while some lane of the wavefront is at or below its LCL; a lane past its LCL is one
        // level behind its loads (the level that crossed waits in wP, wT, wM)
        int k = h.first == DEAD ? nlev + 1 : h.first;
        GPtr lp_, lt_, ld_;
        {
            const int64_t o = (int64_t)(k < nlev ? k : 0) * row_step + lane_off;
            lp_ = (GPtr)b.p.data + o; lt_ = (GPtr)b.t.data + o; ld_ = (GPtr)b.td.data + o;
        }
        double np_ = qnan(), nt_ = qnan(), ntd_ = qnan();
        auto load3 = [&]() __attribute__((always_inline)) {
            np_ = (double)*(GT)lp_; nt_ = (double)*(GT)lt_; ntd_ = (double)*(GT)ld_;
            lp_ += row_step; lt_ += row_step; ld_ += row_step;
            asm volatile("" : "+v"(lp_), "+v"(lt_), "+v"(ld_));
        };
        if (k < nlev) load3();
        double wP = qnan(), wT = qnan(), wM = qnan();
        for (; k <= nlev; ++k) {
            if (__ballot(!h.done) == 0ull) break;
            const bool in = k < nlev;
            const double P = in ? np_ : qnan(), T_ = in ? nt_ : qnan(), M_ = in ? ntd_ : qnan();
            if (k + 1 < nlev) load3();
            const bool skew = h.done;
            if (!skew || k > h.first) feed(h, skew ? wP : P, skew ? wT : T_, skew ? wM : M_, skew, !in);
            wP = P; wT = T_; wM = M_;
        }
        // the next level this chain takes is the one that is waiting (k - 1), or its first one if it has not loaded any:
        // the shared walk feeds level j to a chain in iteration j + 1
        h.first = h.first == DEAD ? DEAD : (k > h.first ? k : h.first + 1);
    });

    // ---- the shared walk: every chain of every lane is above its LCL (k_cape_cin's phase B) ------------------------------
    // h.first now is the iteration in which the chain resumes; the wavefront walks up from the lowest of them, every level
    // is loaded once (one coalesced row request per array) and its environment node -- ln p, Tv(T, Td, p): two e_s behind
    // one wave-uniform range test -- evaluated once for all chains.
    int fmin = h0.first;
    if constexpr (NP > 1) fmin = h1.first < fmin ? h1.first : fmin;
    if constexpr (NP > 2) fmin = h2.first < fmin ? h2.first : fmin;
    int ku = nlev + 1;
    for (int probe = 1; probe <= nlev; ++probe) if (__ballot(fmin <= probe) != 0ull) { ku = probe - 1; break; }
    GPtr lp_, lt_, ld_;
    {
        const int64_t o = (int64_t)(ku < nlev ? ku : 0) * row_step + lane_off;
        lp_ = (GPtr)b.p.data + o; lt_ = (GPtr)b.t.data + o; ld_ = (GPtr)b.td.data + o;
    }
    double np_ = qnan(), nt_ = qnan(), ntd_ = qnan();
    auto load3 = [&]() __attribute__((always_inline)) {
        np_ = (double)*(GT)lp_; nt_ = (double)*(GT)lt_; ntd_ = (double)*(GT)ld_;
        lp_ += row_step; lt_ += row_step; ld_ += row_step;
        asm volatile("" : "+v"(lp_), "+v"(lt_), "+v"(ld_));
    };
    if (ku < nlev) load3();
    Lev cur, prev;
    prev.P = prev.X = prev.T = prev.Td = prev.tve = qnan();
    for (int k = ku; k <= nlev; ++k) {
        each([&](Chain &h, auto ic) __attribute__((always_inline)) {
            if (k >= h.first) {
                const double tvp = h.fam.at(prev.X);
                const double tp = !vtc ? Family::temperature_of(es, prev.P, tvp) : tvp;
                h.sc.template node<false, true>(prev.P, prev.X, vtc ? tvp : tp, vtc ? prev.tve : prev.T, false);
            }
        });
        if (k < nlev) {
            cur.P = np_; cur.T = nt_; cur.Td = ntd_;
            if (k + 1 < nlev) load3();
            cur.X = log_tab<true>(es, cur.P);
            cur.tve = cur.T;
            if (need_w) {
                if (__builtin_amdgcn_ballot_w64(!(in_table(cur.T, 0.0) && in_table(cur.Td, 0.0))) == 0ull) cur.tve = virt_env_tab(es, cur.T, cur.Td, cur.P, true);
                else { double tq = cur.T; asm volatile("" : "+v"(tq)); cur.tve = virt_env_tab(es, tq, cur.Td, cur.P, false); }
            }
            prev = cur;
        }
    }

    // ---- results -------------------------------------------------------------------------------------------------------
    // (output pointers fetched from the kernel arguments only now, as in k_cape_cin: not carried across the walk)
    typedef const MultiArgs __attribute__((address_space(4))) *KernargPtr;
    KernargPtr late = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(late) : : "memory");
    const bool post_zero = late->base.post_zero != 0;
    each([&](Chain &h, auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if (h.first != DEAD) {
            const int of64 = late->s[i].f64;
            Scan::Result r = h.sc.finish(post_zero);
            const int status = h.status | r.status;
            late->flags[i][c] = h.fam.bad ? 1 : 0;
            st(late->s[i].cape, of64, c, r.cape); st(late->s[i].cin, of64, c, r.cin);
            st(late->s[i].lfc_p, of64, c, r.lfc_p); st(late->s[i].lfc_t, of64, c, r.lfc_t);
            st(late->s[i].el_p, of64, c, r.el_p); st(late->s[i].el_t, of64, c, r.el_t);
            sti(late->s[i].lfc_idx, c, r.lfc_idx); sti(late->s[i].el_idx, c, r.el_idx); sti(late->s[i].status, c, status);
        }
    });
    };   // column

    if (PERSIST) {
        const int64_t ntiles = (a.base.ncol + 63) >> 6;
        const int t0 = (int)(ntiles * blockIdx.x / gridDim.x), t1 = (int)(ntiles * (blockIdx.x + 1) / gridDim.x);
        int tile = t0 + (int)(threadIdx.x >> 6);
        while (tile < t1) {
            const int64_t c = ((int64_t)tile << 6) + (threadIdx.x & 63);
            if (c < a.base.ncol) column(c);
            if ((threadIdx.x & 63) == 0) tile = t0 + atomicAdd(&s_next, 1);
            tile = __builtin_amdgcn_readfirstlane(tile);
        }
    } else {
        column(c0);
    }
}

// one translation unit per (T, NP): xp_multi_tu.hip
template <typename T, int NP> void launch_cape_multi(const MultiArgs &a, hipStream_t s);

}  // namespace xp
