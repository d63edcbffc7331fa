// xparcel.hip -- C ABI of libxparcel (include/xparcel.h): argument checking, host<->device staging,
// kernel dispatch.  Everything numerical lives in xp_device.hpp / xp_kernels.hpp.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/xparcel.h"
#include "xp_kernels.hpp"
#include "xp_multi.hpp"
#include "xp_bundle.hpp"

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(XP_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

struct State {
    std::mutex mu;
    bool init = false;
    int device = -1;
    bool tables = false;
    xp::Tables tb{};
    void *tb_index = nullptr, *tb_adiabats = nullptr;
    double *es_tab = nullptr;   // device copy of the e_s(T) polynomial table
    double *fam_tab = nullptr;  // device copy of the adiabat-family table
    std::vector<double> fam_host;
} g;

size_t esize(int dtype) { return dtype == XP_F64 ? 8 : 4; }

// e_s(T) table for xp::es_tab: per 1 K interval the degree-ES_DEG interpolant of Bolton's formula at Chebyshev nodes,
// monomial coefficients in r = T - left edge (what v_fract_f64 delivers), built in long double; layout [coefficient][interval].
void build_es_table(double *out) {
    using LD = long double;
    const int n = xp::ES_DEG + 1;
    const LD pi = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < xp::ES_N; ++i) {
        LD edge = (LD)xp::ES_T_LO + (LD)i;
        LD A[8][9];
        for (int k = 0; k < n; ++k) {
            LD r = 0.5L + 0.5L * cosl(pi * ((LD)k + 0.5L) / (LD)n);
            LD t = edge + r;
            LD v = 1.0L;
            for (int j = 0; j < n; ++j) { A[k][j] = v; v *= r; }
            A[k][n] = 6.112L * expl(17.67L * (t - 273.15L) / (t - 29.65L));
        }
        for (int col = 0; col < n; ++col) {                     // Gaussian elimination with partial pivoting
            int piv = col;
            for (int r = col + 1; r < n; ++r) if (fabsl(A[r][col]) > fabsl(A[piv][col])) piv = r;
            for (int j = 0; j <= n; ++j) { LD t_ = A[col][j]; A[col][j] = A[piv][j]; A[piv][j] = t_; }
            for (int r = 0; r < n; ++r) {
                if (r == col) continue;
                LD f = A[r][col] / A[col][col];
                for (int j = col; j <= n; ++j) A[r][j] -= f * A[col][j];
            }
        }
        for (int j = 0; j < n; ++j) out[j * xp::ES_STRIDE + i] = (double)(A[j][n] / A[j][j]);
    }
    // ln table for xp::log_tab: mantissa interval i of [0.5, 1) has centre c_i = (i + 64.5) / 128
    double *lt = out + xp::LOG_OFF;                    // spare columns of rows 0 (1/c_i) and 1 (ln c_i)
    for (int i = 0; i < xp::LOG_N; ++i) {
        LD c = ((LD)i + 64.5L) / 128.0L;
        lt[i] = (double)(1.0L / c);
        lt[xp::ES_STRIDE + i] = (double)logl(c);
    }
    // exp(i / 64) for xp::dry_factor: spare columns of row 2 (exp(0) = 1 exactly)
    for (int i = 0; i < xp::EXPT_N; ++i) out[xp::EXPT_OFF + i] = (double)expl((LD)(i + xp::EXPT_LO) / 64.0L);
}

// Stages host buffers through device scratch for one call; device buffers pass through.
struct Stager {
    hipStream_t s;
    struct Back { void *host; void *dev; size_t bytes; };
    std::vector<void *> scratch;
    std::vector<Back> back;
    bool any_host = false;
    explicit Stager(void *stream) : s((hipStream_t)stream) {}
    // stream-ordered allocations: a host-array call stages ~25 buffers, and hipMalloc / hipFree would each synchronise
    ~Stager() { for (void *p : scratch) (void)hipFreeAsync(p, s); }
    int in(const void *p, size_t bytes, int mem, const void **out) {
        *out = p;
        if (p == nullptr || mem == XP_MEM_DEVICE) return 0;
        void *d = nullptr;
        HIP_TRY(hipMallocAsync(&d, bytes ? bytes : 1, s));
        scratch.push_back(d);
        HIP_TRY(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s));
        any_host = true;
        *out = d;
        return 0;
    }
    int out(void *p, size_t bytes, int mem, void **dev) {
        *dev = p;
        if (p == nullptr || mem == XP_MEM_DEVICE) return 0;
        void *d = nullptr;
        HIP_TRY(hipMallocAsync(&d, bytes ? bytes : 1, s));
        scratch.push_back(d);
        back.push_back({p, d, bytes});
        any_host = true;
        *dev = d;
        return 0;
    }
    int finish() {
        HIP_TRY(hipGetLastError());
        for (const Back &b : back) HIP_TRY(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, s));
        if (any_host) HIP_TRY(hipStreamSynchronize(s));
        return 0;
    }
};

// adiabat-family table (xp::Family; specification restated independently in oracle/family.py): the VIRTUAL temperature
// T (1 + 0.608 w_s(p, T)) of the parcel along the pseudo-adiabat through every psi-node (Chebyshev points of every psi-piece) is marched from 1000 hPa through the x-nodes (Chebyshev
// points of every x-piece, in order of distance) by classical RK4 with steps <= 1/80, and every (x-piece, psi-piece)
// block of 9 x 9 values is turned into the monomial coefficients of its interpolant (long double elimination).
double fam_dt_dlnp(double x, double t) {
    double p = std::exp(x), e = 6.112 * std::exp(17.67 * (t - 273.15) / (t - 29.65)), pe = p - e;
    double num = xp::RD * t * pe + xp::LV * xp::EPS * e;
    double den = xp::CP_D * xp::RD * t * t * pe + xp::LV * xp::LV * xp::EPS * xp::EPS * e;
    return xp::RD * t * t * num / den;
}
double fam_march(double x, double t, double x1) {
    int n = (int)std::ceil(std::fabs(x1 - x) / 0.0125 - 1e-12);
    if (n < 1) n = 1;
    const double h = (x1 - x) / n;
    for (int s = 0; s < n; ++s) {
        double k1 = fam_dt_dlnp(x, t);
        double k2 = fam_dt_dlnp(x + 0.5 * h, t + 0.5 * h * k1);
        double k3 = fam_dt_dlnp(x + 0.5 * h, t + 0.5 * h * k2);
        double k4 = fam_dt_dlnp(x + h, t + h * k3);
        t = t + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
        x = x + h;
    }
    return t;
}
// monomial coefficients of the polynomial through (u_k, y_k), k < n
void fam_interpolant(int n, const double *u, const double *y, double *c) {
    using LD = long double;
    std::vector<LD> A((size_t)n * (n + 1));
    auto at = [&](int r, int col) -> LD & { return A[(size_t)r * (n + 1) + col]; };
    for (int k = 0; k < n; ++k) {
        LD v = 1.0L;
        for (int i = 0; i < n; ++i) { at(k, i) = v; v *= (LD)u[k]; }
        at(k, n) = (LD)y[k];
    }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r) if (fabsl(at(r, col)) > fabsl(at(piv, col))) piv = r;
        for (int i = 0; i <= n; ++i) std::swap(at(col, i), at(piv, i));
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            LD f = at(r, col) / at(col, col);
            for (int i = col; i <= n; ++i) at(r, i) -= f * at(col, i);
        }
    }
    for (int i = 0; i < n; ++i) c[i] = (double)(at(i, n) / at(i, i));
}
const double kFamEdges[xp::FAM_NPS + 1] = XP_FAM_EDGES;
// the psi-piece centres and reciprocal half-widths the device reads behind the coefficients
void fam_append_pieces(double *tab) {
    for (int q = 0; q < xp::FAM_NPS; ++q) {
        tab[xp::FAM_COEFS + q] = 0.5 * (kFamEdges[q] + kFamEdges[q + 1]);
        tab[xp::FAM_COEFS + xp::FAM_NPS + q] = 1.0 / (0.5 * (kFamEdges[q + 1] - kFamEdges[q]));
    }
}
// The device copy is laid out [power of s][power of z][x-piece][psi-piece] (+ the appended piece constants): the nine
// coefficients a lane multiplies through one Horner row are then 5184 B apart, beyond the reach of ds_read2_b64, so the
// compiler issues plain ds_read_b64 (2 LDS cycles each, banks mod 64) instead of pairing them (8 cycles per pair, banks mod
// 32) -- the same remedy as the e_s table's row stride (xp_device.hpp).  The ABI / oracle order stays [x-piece][z][s][psi].
std::vector<double> family_device_layout(const std::vector<double> &host) {
    const int NN = xp::FAM_ND + 1, MM = xp::FAM_MD + 1;
    std::vector<double> dev(host.size());
    for (int j = 0; j < xp::FAM_NPX; ++j)
        for (int n = 0; n < NN; ++n)
            for (int m = 0; m < MM; ++m)
                for (int q = 0; q < xp::FAM_NPS; ++q)
                    dev[(((size_t)m * NN + n) * xp::FAM_NPX + j) * xp::FAM_NPS + q] = host[(((size_t)j * NN + n) * MM + m) * xp::FAM_NPS + q];
    for (int i = xp::FAM_COEFS; i < xp::FAM_SIZE; ++i) dev[i] = host[i];
    return dev;
}
void build_family_table(double *tab) {
    const int NN = xp::FAM_ND + 1, MM = xp::FAM_MD + 1, NXN = xp::FAM_NPX * NN, NSN = xp::FAM_NPS * MM;
    const double pi = 3.14159265358979323846;
    std::vector<double> un(NN), um(MM), xs(NXN), ps(NSN), vals((size_t)NXN * NSN);
    for (int k = 0; k < NN; ++k) un[k] = std::cos(pi * (k + 0.5) / NN);
    for (int k = 0; k < MM; ++k) um[k] = std::cos(pi * (k + 0.5) / MM);
    for (int j = 0; j < xp::FAM_NPX; ++j)
        for (int k = 0; k < NN; ++k) xs[j * NN + k] = (xp::FAM_XHI - xp::FAM_WX * (j + 0.5)) + 0.5 * xp::FAM_WX * un[k];
    for (int q = 0; q < xp::FAM_NPS; ++q)
        for (int k = 0; k < MM; ++k)
            ps[q * MM + k] = 0.5 * (kFamEdges[q] + kFamEdges[q + 1]) + 0.5 * (kFamEdges[q + 1] - kFamEdges[q]) * um[k];
    for (int side = 0; side < 2; ++side) {
        std::vector<int> order;
        for (int i = 0; i < NXN; ++i) if ((side == 0) == (xs[i] <= xp::FAM_X1000)) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return std::fabs(xs[a] - xp::FAM_X1000) < std::fabs(xs[b] - xp::FAM_X1000); });
        for (int c = 0; c < NSN; ++c) {
            double x = xp::FAM_X1000, t = ps[c];
            for (int i : order) {
                if (xs[i] != x) { t = fam_march(x, t, xs[i]); x = xs[i]; }
                // the parcel's virtual temperature along the adiabat (pf.py:760 + 775), which is what the table holds
                const double pr = std::exp(xs[i]), e = 6.112 * std::exp(17.67 * (t - 273.15) / (t - 29.65));
                vals[(size_t)i * NSN + c] = t * (1.0 + xp::VT_EPS * (xp::EPS * e / (pr - e)));
            }
        }
    }
    std::vector<double> a((size_t)NN * MM), col(NN), cf(NN), row(MM), rf(MM);
    for (int j = 0; j < xp::FAM_NPX; ++j)
        for (int q = 0; q < xp::FAM_NPS; ++q) {
            for (int m = 0; m < MM; ++m) {
                for (int k = 0; k < NN; ++k) col[k] = vals[(size_t)(j * NN + k) * NSN + (q * MM + m)];
                fam_interpolant(NN, un.data(), col.data(), cf.data());
                for (int n = 0; n < NN; ++n) a[(size_t)n * MM + m] = cf[n];
            }
            for (int n = 0; n < NN; ++n) {
                for (int m = 0; m < MM; ++m) row[m] = a[(size_t)n * MM + m];
                fam_interpolant(MM, um.data(), row.data(), rf.data());
                for (int m = 0; m < MM; ++m) tab[(((size_t)j * NN + n) * MM + m) * xp::FAM_NPS + q] = rf[m];
            }
        }
    fam_append_pieces(tab);
}

int check_view(const xp_view *v, const char *name) {
    if (!v || !v->data) return fail(XP_E_ARG, "%s: null view", name);
    if (v->dtype != XP_F32 && v->dtype != XP_F64) return fail(XP_E_ARG, "%s: dtype must be XP_F32 or XP_F64", name);
    if (v->nlev < 1 || v->ncol < 0 || v->nlev >= (1ll << 30))
        return fail(XP_E_ARG, "%s: bad shape (%lld, %lld)", name, (long long)v->nlev, (long long)v->ncol);
    if (v->mem == XP_MEM_HOST && !(v->col_stride == 1 && v->lev_stride == v->ncol))
        return fail(XP_E_ARG, "%s: host views must be dense (nlev, ncol) C-order", name);
    return 0;
}
int same_shape(const xp_view *a, const xp_view *b, const char *what) {
    if (a->nlev != b->nlev || a->ncol != b->ncol || a->dtype != b->dtype)
        return fail(XP_E_ARG, "%s: views differ in shape or dtype", what);
    return 0;
}
int stage_view(Stager &st, const xp_view *v, xp::View *out) {
    const void *d;
    int rc = st.in(v->data, (size_t)v->nlev * (size_t)v->ncol * esize(v->dtype), v->mem, &d);
    if (rc) return rc;
    out->data = d; out->ls = v->lev_stride; out->cs = v->col_stride;
    return 0;
}
// every entry point switches to the library's device for its duration and leaves the calling thread's current device
// as it found it (the caller -- torch, say -- may be working on another one)
struct DevGuard {
    int prev = -1;
    DevGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
int ensure_init() {
    if (!g.init) return fail(XP_E_NOT_INIT, "xp_init() has not been called");
    hipError_t e = hipSetDevice(g.device);
    if (e != hipSuccess) return fail(XP_E_HIP, "hipSetDevice(%d): %s", g.device, hipGetErrorString(e));
    return 0;
}
unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

int stage_scalars(Stager &st, xp_scalars_out *s, int64_t ncol, xp::ScalarsOut *o) {
    memset(o, 0, sizeof(*o));
    if (!s) { o->f64 = 1; return 0; }
    if (s->dtype != XP_F32 && s->dtype != XP_F64) return fail(XP_E_ARG, "scalars: bad dtype");
    o->f64 = s->dtype == XP_F64;
    size_t fb = (size_t)ncol * esize(s->dtype), ib = (size_t)ncol * 4;
    int rc = 0;
#define F_(dst, src) if (!rc) rc = st.out(s->src, fb, s->mem, &o->dst)
#define I_(dst, src) if (!rc) { void *t_; rc = st.out(s->src, ib, s->mem, &t_); o->dst = (int32_t *)t_; }
    F_(cape, cape); F_(cin, cin); F_(lcl_p, lcl_pressure); F_(lcl_t, lcl_temperature); F_(lcl_tv, lcl_virtual_temperature);
    F_(lfc_p, lfc_pressure); F_(lfc_t, lfc_temperature); F_(el_p, el_pressure); F_(el_t, el_temperature);
    I_(lfc_idx, lfc_index); I_(el_idx, el_index); I_(status, status); I_(parcel_idx, parcel_index);
    F_(par_p, parcel_pressure); F_(par_t, parcel_temperature); F_(par_td, parcel_dewpoint);
#undef F_
#undef I_
    return rc;
}

// k_cape_cin's 96 instantiations are compiled in six translation units (xp_cape_tu.hip, one per dtype x moist mode) so
// that the build parallelises; this file only dispatches to them.
template <typename T, int PM> void launch_cape(const xp::CapeArgs &a, bool profile, hipStream_t s) {
    if (a.ncol == 0) return;
    if (a.table_mode) xp::launch_cape_mode<T, 1>(a, PM, profile, s);
    else if (a.flags) {                                      // family mode: fast pass, then RK4 for the flagged columns
        xp::launch_cape_mode<T, 2>(a, PM, profile, s);
        xp::CapeArgs b = a;
        b.only_flagged = 1;
        xp::launch_cape_mode<T, 0>(b, PM, profile, s);
    } else xp::launch_cape_mode<T, 0>(a, PM, profile, s);
}
template <typename T> void launch_cape_pm(const xp::CapeArgs &a, int pm, bool profile, hipStream_t s) {
    switch (pm) {
        case XP_PARCEL_SURFACE: launch_cape<T, xp::PM_SURFACE>(a, profile, s); break;
        case XP_PARCEL_MOST_UNSTABLE: launch_cape<T, xp::PM_MU>(a, profile, s); break;
        case XP_PARCEL_MIXED_LAYER: launch_cape<T, xp::PM_ML>(a, profile, s); break;
        default: launch_cape<T, xp::PM_EXPLICIT>(a, profile, s); break;
    }
}

int fill_common(Stager &st, const xp_view *p, const xp_view *t, const xp_view *td, const xp_parcel *parcel,
                const xp_opts *o, xp::CapeArgs *a) {
    int rc;
    if ((rc = check_view(p, "pressure")) || (rc = check_view(t, "temperature")) || (rc = check_view(td, "dewpoint"))) return rc;
    if ((rc = same_shape(p, t, "pressure/temperature")) || (rc = same_shape(p, td, "pressure/dewpoint"))) return rc;
    if (!parcel) return fail(XP_E_ARG, "parcel: null");
    if (parcel->mode < XP_PARCEL_SURFACE || parcel->mode > XP_PARCEL_EXPLICIT) return fail(XP_E_ARG, "parcel: bad mode");
    memset(a, 0, sizeof(*a));
    if ((rc = stage_view(st, p, &a->p)) || (rc = stage_view(st, t, &a->t)) || (rc = stage_view(st, td, &a->td))) return rc;
    a->nlev = p->nlev; a->ncol = p->ncol;
    a->depth = parcel->depth;
    {   // the library's tables: read under the lock that xp_init / xp_set_tables / xp_set_family_table replace them under
        std::lock_guard<std::mutex> lk(g.mu);
        a->es_tab = g.es_tab;
        a->fam_tab = g.fam_tab;
        a->tb = g.tb;
    }
    {   // fast level addressing (xp_kernels.hpp load3): common strides, non-negative, column offsets below 4 GiB
        bool same = a->p.ls == a->t.ls && a->p.ls == a->td.ls && a->p.cs == a->t.cs && a->p.cs == a->td.cs;
        bool fits = a->p.cs >= 0 && a->p.ls >= 0 &&
                    (unsigned long long)a->p.cs * (unsigned long long)(p->ncol > 0 ? p->ncol : 1) * esize(p->dtype) < (1ull << 32);
        a->off32 = 1;
        if (!(same && fits)) {                               // rare: densify the three views on the device first
            size_t n = (size_t)p->nlev * (size_t)p->ncol, bytes = n * esize(p->dtype);
            xp::View *vs[3] = {&a->p, &a->t, &a->td};
            for (int i = 0; i < 3 && n; ++i) {
                void *d = nullptr;
                HIP_TRY(hipMallocAsync(&d, bytes, st.s));
                st.scratch.push_back(d);
                if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_densify<double>), dim3(blocks((int64_t)n)), dim3(256), 0, st.s, *vs[i], p->nlev, p->ncol, (double *)d);
                else hipLaunchKernelGGL((xp::k_densify<float>), dim3(blocks((int64_t)n)), dim3(256), 0, st.s, *vs[i], p->nlev, p->ncol, (float *)d);
                vs[i]->data = d; vs[i]->ls = p->ncol; vs[i]->cs = 1;
            }
            if ((unsigned long long)(p->ncol > 0 ? p->ncol : 1) * esize(p->dtype) >= (1ull << 32))
                return fail(XP_E_ARG, "more than 4 GiB per level row");
        }
    }
    if (parcel->mode == XP_PARCEL_EXPLICIT) {
        if (!parcel->pressure || !parcel->temperature || !parcel->dewpoint) return fail(XP_E_ARG, "explicit parcel: null arrays");
        size_t b = (size_t)p->ncol * esize(p->dtype);
        if ((rc = st.in(parcel->pressure, b, p->mem, &a->ex_p)) || (rc = st.in(parcel->temperature, b, p->mem, &a->ex_t)) ||
            (rc = st.in(parcel->dewpoint, b, p->mem, &a->ex_td))) return rc;
    }
    if (o) {
        if (o->lcl_interp != XP_LCL_INTERP_LINEAR && o->lcl_interp != XP_LCL_INTERP_LOG)
            return fail(XP_E_INTERP, "interpolator must be linear or log");
        if (o->moist_mode != XP_MOIST_EXACT && o->moist_mode != XP_MOIST_TABLE && o->moist_mode != XP_MOIST_FAMILY)
            return fail(XP_E_ARG, "bad moist_mode");
        if (o->compute != XP_F64) return fail(XP_E_ARG, "xp_opts.compute: only XP_F64 arithmetic is implemented");
        a->vtc = o->virtual_temperature_correction; a->log_interp = o->lcl_interp == XP_LCL_INTERP_LOG;
        if (o->humidity != XP_HUM_DEWPOINT && o->humidity != XP_HUM_SPECIFIC) return fail(XP_E_ARG, "bad xp_opts.humidity");
        a->pos_neg = o->pos_cape_neg_cin; a->post_zero = o->post_zero_cin; a->table_mode = o->moist_mode == XP_MOIST_TABLE;
        a->hum = o->humidity == XP_HUM_SPECIFIC;
    } else {
        a->vtc = 1; a->log_interp = 1; a->pos_neg = 1; a->post_zero = 0; a->table_mode = 0; a->hum = 0;
    }
    if (a->table_mode) {
        if (!g.tables) return fail(XP_E_NO_TABLES, "Call load_moist_adiabat_lookups first.");
    }
    return 0;
}

}  // namespace

namespace {
template <typename T, int NV> void launch_interp_levels(int nt, const xp::View &cv, const xp::InterpMany &m, int64_t nlev, int64_t ncol, int lg, hipStream_t s) {
    dim3 gr(blocks(ncol)), bl(256);
    switch (nt) {
        case 1: hipLaunchKernelGGL((xp::k_interp_levels<T, NV, 1>), gr, bl, 0, s, cv, m, nlev, ncol, lg); break;
        case 2: hipLaunchKernelGGL((xp::k_interp_levels<T, NV, 2>), gr, bl, 0, s, cv, m, nlev, ncol, lg); break;
        case 3: hipLaunchKernelGGL((xp::k_interp_levels<T, NV, 3>), gr, bl, 0, s, cv, m, nlev, ncol, lg); break;
        default: hipLaunchKernelGGL((xp::k_interp_levels<T, NV, 4>), gr, bl, 0, s, cv, m, nlev, ncol, lg); break;
    }
}
template <typename T> void launch_interp_levels_v(int nv, int nt, const xp::View &cv, const xp::InterpMany &m, int64_t nlev, int64_t ncol, int lg, hipStream_t s) {
    switch (nv) {
        case 1: launch_interp_levels<T, 1>(nt, cv, m, nlev, ncol, lg, s); break;
        case 2: launch_interp_levels<T, 2>(nt, cv, m, nlev, ncol, lg, s); break;
        case 3: launch_interp_levels<T, 3>(nt, cv, m, nlev, ncol, lg, s); break;
        default: launch_interp_levels<T, 4>(nt, cv, m, nlev, ncol, lg, s); break;
    }
}
}  // namespace

extern "C" {

int xp_version(void) { return XP_VERSION; }
const char *xp_last_error(void) { return g_err; }

int xp_init(int device) {
    std::lock_guard<std::mutex> lk(g.mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(XP_E_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(XP_E_ARG, "device %d out of range (0..%d)", device, n - 1);
    DevGuard dg_;
    if (g.init && g.device != device) {                      // moving to another device: nothing queued on the old one may
        HIP_TRY(hipSetDevice(g.device));                     // still be reading the buffers freed below
        HIP_TRY(hipDeviceSynchronize());
    }
    HIP_TRY(hipSetDevice(device));
    if (g.init && g.device != device && g.tables) {
        // tables live on the old device; drop them, the caller reloads
        (void)hipFree(g.tb_index); (void)hipFree(g.tb_adiabats);
        g.tb_index = g.tb_adiabats = nullptr; g.tables = false;
    }
    if (g.init && g.device != device && g.es_tab) { (void)hipFree(g.es_tab); g.es_tab = nullptr; }
    if (!g.es_tab) {
        std::vector<double> tab(xp::LDS_TAB);
        build_es_table(tab.data());
        HIP_TRY(hipMalloc((void **)&g.es_tab, sizeof(double) * xp::LDS_TAB));
        HIP_TRY(hipMemcpy(g.es_tab, tab.data(), sizeof(double) * xp::LDS_TAB, hipMemcpyHostToDevice));
    }
    if (g.init && g.device != device && g.fam_tab) { (void)hipFree(g.fam_tab); g.fam_tab = nullptr; }
    if (!g.fam_tab) {
        if (g.fam_host.empty()) { g.fam_host.resize((size_t)xp::FAM_SIZE); build_family_table(g.fam_host.data()); }
        HIP_TRY(hipMalloc((void **)&g.fam_tab, sizeof(double) * g.fam_host.size()));
        HIP_TRY(hipMemcpy(g.fam_tab, family_device_layout(g.fam_host).data(), sizeof(double) * g.fam_host.size(), hipMemcpyHostToDevice));
    }
    g.device = device;
    g.init = true;
    return XP_OK;
}

int xp_set_tables(const xp_tables *t) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    if (!t || !t->index || !t->adiabats || t->n_pressure < 2 || t->n_temperature < 2 || t->n_adiabat < 1)
        return fail(XP_E_ARG, "xp_set_tables: bad tables");
    if (g.tables) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(g.tb_index); (void)hipFree(g.tb_adiabats); g.tables = false; }
    {   // an index entry beyond n_adiabat would send Moist::start outside the adiabat array
        const uint16_t *ix = (const uint16_t *)t->index;
        size_t n_ix = (size_t)t->n_pressure * (size_t)t->n_temperature;
        for (size_t i = 0; i < n_ix; ++i)
            if ((int64_t)ix[i] > t->n_adiabat) return fail(XP_E_ARG, "xp_set_tables: index entry %u > n_adiabat %lld", (unsigned)ix[i], (long long)t->n_adiabat);
    }
    size_t ib = (size_t)t->n_pressure * (size_t)t->n_temperature * 2, ab = (size_t)t->n_adiabat * (size_t)t->n_pressure * 4;
    HIP_TRY(hipMalloc(&g.tb_index, ib));
    HIP_TRY(hipMalloc(&g.tb_adiabats, ab));
    HIP_TRY(hipMemcpy(g.tb_index, t->index, ib, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(g.tb_adiabats, t->adiabats, ab, hipMemcpyHostToDevice));
    g.tb.index = (const uint16_t *)g.tb_index; g.tb.adiabats = (const float *)g.tb_adiabats;
    g.tb.n_p = t->n_pressure; g.tb.n_t = t->n_temperature;
    g.tb.p_max = t->p_max; g.tb.p_step = t->p_step; g.tb.t_min = t->t_min; g.tb.t_step = t->t_step;
    g.tables = true;
    return XP_OK;
}
int xp_tables_loaded(void) { return g.tables ? 1 : 0; }

int xp_family_table(double *out, int64_t *n_lnp, int64_t *n_label) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (n_lnp) *n_lnp = xp::FAM_COEFS / xp::FAM_NPS;
    if (n_label) *n_label = xp::FAM_NPS;
    if (out) memcpy(out, g.fam_host.data(), sizeof(double) * xp::FAM_COEFS);
    return XP_OK;
}
int xp_set_family_table(const double *tab, int64_t n_lnp, int64_t n_label) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (!tab || n_lnp != xp::FAM_COEFS / xp::FAM_NPS || n_label != xp::FAM_NPS) return fail(XP_E_ARG, "xp_set_family_table: wrong shape");
    std::lock_guard<std::mutex> lk(g.mu);
    HIP_TRY(hipDeviceSynchronize());                         // kernels in flight may still read the old table
    memcpy(g.fam_host.data(), tab, sizeof(double) * xp::FAM_COEFS);
    HIP_TRY(hipMemcpy(g.fam_tab, family_device_layout(g.fam_host).data(), sizeof(double) * g.fam_host.size(), hipMemcpyHostToDevice));
    return XP_OK;
}

int xp_cape_cin(const xp_view *p, const xp_view *t, const xp_view *td, const xp_parcel *parcel, const xp_opts *o,
                xp_scalars_out *scalars, xp_profile_out *profile, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    Stager st(stream);
    xp::CapeArgs a;
    if ((rc = fill_common(st, p, t, td, parcel, o, &a))) return rc;
    if ((rc = stage_scalars(st, scalars, a.ncol, &a.s))) return rc;
    if (profile) {
        if (profile->dtype != XP_F32 && profile->dtype != XP_F64) return fail(XP_E_ARG, "profile: bad dtype");
        if (profile->nlev_out < a.nlev + 1) return fail(XP_E_ARG, "profile: nlev_out must be >= nlev + 1");
        if (profile->mem == XP_MEM_HOST && !(profile->col_stride == 1 && profile->lev_stride == a.ncol))
            return fail(XP_E_ARG, "profile: host arrays must be dense (nlev_out, ncol)");
        void *src[6] = {profile->pressure, profile->temperature, profile->virtual_temperature,
                        profile->environment_temperature, profile->environment_virtual_temperature,
                        profile->environment_dewpoint};
        size_t b = (size_t)profile->nlev_out * (size_t)a.ncol * esize(profile->dtype);
        for (int v = 0; v < 6; ++v)
            if ((rc = st.out(src[v], b, profile->mem, &a.prof.v[v]))) return rc;
        a.prof.nlev_out = profile->nlev_out; a.prof.ls = profile->lev_stride; a.prof.cs = profile->col_stride;
        bool rows = false;
        for (int v = 0; v < 6; ++v) rows = rows || a.prof.v[v] != nullptr;
        if (!rows) a.prof.nlev_out = 0;                    // lifted index only: the kernel's row loops see an empty profile
        a.prof.f64 = profile->dtype == XP_F64;
        bool all6 = true;
        for (int v = 0; v < 6; ++v) all6 = all6 && a.prof.v[v] != nullptr;
        a.prof.native6 = all6 && profile->dtype == p->dtype;
        if (profile->lifted_index) {
            if (!(profile->lifted_index_pressure > 0.0)) return fail(XP_E_ARG, "profile: lifted_index_pressure must be positive");
            if ((rc = st.out(profile->lifted_index, (size_t)a.ncol * esize(profile->dtype), profile->mem, &a.prof.li))) return rc;
            a.prof.li_x = log(profile->lifted_index_pressure);
        }
    }
    void *flags = nullptr;
    if (o && o->moist_mode == XP_MOIST_FAMILY && a.ncol > 0) {
        HIP_TRY(hipMallocAsync(&flags, sizeof(int32_t) * (size_t)a.ncol, st.s));   // stream-ordered scratch: which columns need RK4
        st.scratch.push_back(flags);                         // released by the Stager on every path out of this function
        a.flags = (int32_t *)flags;
        // larger grids run persistent wavefronts (k_cape_cin, PERSIST); XP_PERSIST_MIN_COLS: A/B.  Measured per grid size
        // (profiles/r03_persist.txt, scripts/run_gpu_persist.py): equal to the ordinary launch up to two rounds of
        // workgroups (512 Ki columns), 10-13 % faster from three rounds on (768 Ki ... 2 Mi columns of 64 f64 levels; c2:
        // 0.64 -> 0.575 ms) -- a workgroup's sixteen wavefronts no longer wait for the slowest of them before the next
        // sixteen tiles start.  (Round 2's kernels gained nothing from it below 4 Mi columns.)
        static const long long persist_env = [] { const char *e = getenv("XP_PERSIST_MIN_COLS"); return e ? atoll(e) : -1ll; }();
        const bool searching = parcel->mode == XP_PARCEL_MOST_UNSTABLE || parcel->mode == XP_PARCEL_MIXED_LAYER;
        const long long persist_min = persist_env >= 0 ? persist_env : searching ? (1ll << 19) : (3ll << 18);
        a.persist = (long long)a.ncol >= persist_min && a.ncol < (1ll << 36);
    }
    if (p->dtype == XP_F64) launch_cape_pm<double>(a, parcel->mode, profile != nullptr, st.s);
    else launch_cape_pm<float>(a, parcel->mode, profile != nullptr, st.s);
    return st.finish();
}

namespace {
bool multi_fused_ok(int32_t np, const xp_parcel *parcels, const xp_opts *o, const xp_profile_out *profiles) {
    static const bool force = getenv("XP_MULTI_FUSED") != nullptr;           // A/B: fuse whenever possible
    if (!o || o->moist_mode != XP_MOIST_FAMILY || o->humidity != XP_HUM_DEWPOINT) return false;
    if (!force && !(o->flags & XP_OPT_FUSE_PARCELS)) return false;
    if (np != 2) return false;                                               // instantiated parcel counts (xp_multi_tu.hip)
    for (int i = 0; i < np; ++i) {
        if (parcels[i].mode != XP_PARCEL_SURFACE && parcels[i].mode != XP_PARCEL_MOST_UNSTABLE && parcels[i].mode != XP_PARCEL_MIXED_LAYER) return false;
        if (profiles) return false;
    }
    return true;
}
}  // namespace

int xp_cape_cin_multi(const xp_view *p, const xp_view *t, const xp_view *td, int32_t np, const xp_parcel *parcels,
                      const xp_opts *o, xp_scalars_out *scalars, xp_profile_out *profiles, void *stream) {
    if (np < 1 || np > xp::MULTI_MAX) return fail(XP_E_ARG, "xp_cape_cin_multi: 1...%d parcels", xp::MULTI_MAX);
    if (!parcels || !scalars) return fail(XP_E_ARG, "xp_cape_cin_multi: null parcels / scalars");
    if (!multi_fused_ok(np, parcels, o, profiles)) {
        for (int i = 0; i < np; ++i) {
            int rc = xp_cape_cin(p, t, td, &parcels[i], o, &scalars[i], profiles ? &profiles[i] : nullptr, stream);
            if (rc) return rc;
        }
        return XP_OK;
    }
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    Stager st(stream);
    xp::MultiArgs m;
    memset(&m, 0, sizeof(m));
    if ((rc = fill_common(st, p, t, td, &parcels[0], o, &m.base))) return rc;
    m.np = np;
    const int64_t ncol = m.base.ncol;
    for (int i = 0; i < np; ++i) {
        m.mode[i] = parcels[i].mode;                                         // XP_PARCEL_* == xp::PM_*
        m.depth[i] = parcels[i].depth;
        if ((rc = stage_scalars(st, &scalars[i], ncol, &m.s[i]))) return rc;
    }
    if (ncol == 0) return st.finish();
    void *flags = nullptr;
    HIP_TRY(hipMallocAsync(&flags, sizeof(int32_t) * (size_t)ncol * (size_t)np, st.s));
    st.scratch.push_back(flags);
    for (int i = 0; i < np; ++i) m.flags[i] = (int32_t *)flags + (size_t)i * (size_t)ncol;
    static const long long persist_env = [] { const char *e = getenv("XP_PERSIST_MIN_COLS"); return e ? atoll(e) : -1ll; }();
    m.base.persist = (long long)ncol >= (persist_env >= 0 ? persist_env : (1ll << 19)) && ncol < (1ll << 36);
    const bool f64 = p->dtype == XP_F64;
    if (f64) xp::launch_cape_multi<double, 2>(m, st.s); else xp::launch_cape_multi<float, 2>(m, st.s);
    // columns a chain's family table could not serve: the single-parcel RK4 kernel redoes them, chain by chain
    for (int i = 0; i < np; ++i) {
        xp::CapeArgs b = m.base;
        b.depth = m.depth[i]; b.s = m.s[i]; b.flags = m.flags[i]; b.only_flagged = 1; b.persist = 0;
        if (f64) xp::launch_cape_mode<double, 0>(b, m.mode[i], false, st.s);
        else xp::launch_cape_mode<float, 0>(b, m.mode[i], false, st.s);
    }
    return st.finish();
}

int xp_conv_properties(const xp_conv_in *in, const xp_opts *o, int32_t ignore_nans, xp_conv_out *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (!in || !out) return fail(XP_E_ARG, "xp_conv_properties: null argument");
    const xp_view *p = in->pressure, *t = in->temperature, *q = in->specific_humidity, *z = in->height_asl;
    const xp_view *wu = in->wind_u, *wv = in->wind_v, *wh = in->wind_height_above_surface;
    if ((rc = check_view(p, "pressure")) || (rc = check_view(t, "temperature")) || (rc = check_view(q, "specific_humidity")) ||
        (rc = check_view(z, "height_asl")) || (rc = same_shape(p, t, "pressure/temperature")) ||
        (rc = same_shape(p, q, "pressure/specific_humidity")) || (rc = same_shape(p, z, "pressure/height_asl"))) return rc;
    if ((rc = check_view(wu, "wind_u")) || (rc = check_view(wv, "wind_v")) || (rc = check_view(wh, "wind_height_above_surface")) ||
        (rc = same_shape(wu, wv, "wind_u/wind_v")) || (rc = same_shape(wu, wh, "wind_u/wind_height_above_surface"))) return rc;
    if (wu->ncol != p->ncol || wu->dtype != p->dtype || wu->mem != p->mem || t->mem != p->mem || q->mem != p->mem || z->mem != p->mem)
        return fail(XP_E_ARG, "xp_conv_properties: the views must agree in columns, dtype and mem");
    if (!in->surface_wind_u || !in->surface_wind_v) return fail(XP_E_ARG, "xp_conv_properties: null surface wind");
    Stager st(stream);
    const int64_t nlev = p->nlev, ncol = p->ncol;
    const size_t es = esize(p->dtype), cb = (size_t)ncol * es;
    const int mem = p->mem;
    if (ncol == 0) return XP_OK;
    // inputs on the device (host views are dense by contract: staged as they are)
    xp_view dv[7];
    const xp_view *src[7] = {p, t, q, z, wu, wv, wh};
    for (int i = 0; i < 7; ++i) {
        dv[i] = *src[i];
        const void *d;
        if ((rc = st.in(src[i]->data, (size_t)src[i]->nlev * (size_t)ncol * es, mem, &d))) return rc;
        dv[i].data = d; dv[i].mem = XP_MEM_DEVICE;
    }
    const void *sfu, *sfv;
    if ((rc = st.in(in->surface_wind_u, cb, mem, &sfu)) || (rc = st.in(in->surface_wind_v, cb, mem, &sfv))) return rc;
    // scratch: the dewpoint grid and per-point temporaries
    auto scratch = [&](size_t bytes, void **ptr) -> int {
        HIP_TRY(hipMallocAsync(ptr, bytes ? bytes : 1, st.s));
        st.scratch.push_back(*ptr);
        return 0;
    };
    void *td = nullptr, *tmp = nullptr, *valid = nullptr;
    enum { T850, T700, T500, TD850, Z700, Z500, HIU, HIV, MUP, MUTD, NTMP };
    if ((rc = scratch((size_t)nlev * cb, &td)) || (rc = scratch((size_t)NTMP * cb, &tmp)) || (rc = scratch((size_t)ncol * 4, &valid))) return rc;
    auto tp = [&](int i) { return (void *)((char *)tmp + (size_t)i * cb); };
    // outputs: the caller's buffers when they live on the device, staged otherwise; results that are also inputs of the
    // per-point kernel need a buffer even when the caller does not want them
    void *o_[21];
    void *const want[21] = {out->mu_cape, out->mu_cin, out->mu_mixing_ratio, out->mu_lifted_index, out->mu_dci,
                            out->mixed_100_cape, out->mixed_100_cin, out->mixed_100_lifted_index, out->mixed_100_dci,
                            out->mixed_50_cape, out->mixed_50_cin, out->mixed_50_lifted_index, out->mixed_50_dci,
                            out->lapse_rate_700_500, out->temp_500, out->freezing_level, out->melting_level,
                            out->shear_u, out->shear_v, out->shear_magnitude, (void *)out->positive_shear};
    for (int i = 0; i < 21; ++i) {
        const size_t b = i == 20 ? (size_t)ncol * 4 : cb;
        if (want[i]) { if ((rc = st.out(want[i], b, mem, &o_[i]))) return rc; }
        else if ((rc = scratch(b, &o_[i]))) return rc;
    }
    enum { O_MU_CAPE, O_MU_CIN, O_MU_W, O_MU_LI, O_MU_DCI, O_M1_CAPE, O_M1_CIN, O_M1_LI, O_M1_DCI, O_M5_CAPE, O_M5_CIN, O_M5_LI, O_M5_DCI,
           O_LAPSE, O_T500, O_FRZ, O_MLT, O_SHU, O_SHV, O_SHM, O_POS };
    // 1. one pass over the four grids
    xp::ConvColumnsArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.p = {dv[0].data, dv[0].lev_stride, dv[0].col_stride}; ca.t = {dv[1].data, dv[1].lev_stride, dv[1].col_stride};
    ca.q = {dv[2].data, dv[2].lev_stride, dv[2].col_stride}; ca.z = {dv[3].data, dv[3].lev_stride, dv[3].col_stride};
    ca.nlev = nlev; ca.ncol = ncol; ca.td = td;
    ca.t850 = tp(T850); ca.t700 = tp(T700); ca.t500 = tp(T500); ca.td850 = tp(TD850); ca.z700 = tp(Z700); ca.z500 = tp(Z500);
    ca.freezing = o_[O_FRZ]; ca.melting = o_[O_MLT]; ca.valid = (int32_t *)valid;
    if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_conv_columns<double>), dim3(blocks(ncol)), dim3(256), 0, st.s, ca);
    else hipLaunchKernelGGL((xp::k_conv_columns<float>), dim3(blocks(ncol)), dim3(256), 0, st.s, ca);
    // 2. the three parcels (pf.py:1984-2006), lifted index out of the same passes
    xp_view tdv = dv[0];
    tdv.data = td; tdv.lev_stride = ncol; tdv.col_stride = 1;
    xp_opts oo;
    if (o) oo = *o; else { memset(&oo, 0, sizeof(oo)); oo.virtual_temperature_correction = 1; oo.lcl_interp = XP_LCL_INTERP_LOG; oo.pos_cape_neg_cin = 1; oo.compute = XP_F64; }
    oo.humidity = XP_HUM_DEWPOINT;
    const struct { int mode; double depth; int cape, cin, li; } pc[3] = {{XP_PARCEL_MOST_UNSTABLE, 250.0, O_MU_CAPE, O_MU_CIN, O_MU_LI},
                                                                        {XP_PARCEL_MIXED_LAYER, 100.0, O_M1_CAPE, O_M1_CIN, O_M1_LI},
                                                                        {XP_PARCEL_MIXED_LAYER, 50.0, O_M5_CAPE, O_M5_CIN, O_M5_LI}};
    for (int i = 0; i < 3; ++i) {
        xp_parcel par;
        memset(&par, 0, sizeof(par));
        par.mode = pc[i].mode; par.depth = pc[i].depth;
        xp_scalars_out so;
        memset(&so, 0, sizeof(so));
        so.dtype = p->dtype; so.mem = XP_MEM_DEVICE; so.cape = o_[pc[i].cape]; so.cin = o_[pc[i].cin];
        if (i == 0) { so.parcel_pressure = tp(MUP); so.parcel_dewpoint = tp(MUTD); }
        xp_profile_out po;
        memset(&po, 0, sizeof(po));
        po.dtype = p->dtype; po.mem = XP_MEM_DEVICE; po.nlev_out = nlev + 1; po.lev_stride = ncol; po.col_stride = 1;
        po.lifted_index = o_[pc[i].li]; po.lifted_index_pressure = 500.0;
        if ((rc = xp_cape_cin(&dv[0], &dv[1], &tdv, &par, &oo, &so, &po, stream))) return rc;
    }
    // 3. wind at 6000 m above the surface (pf.py:2240-2243: linear interpolation in height)
    {
        const xp_view *vars[2] = {&dv[4], &dv[5]};
        const double at6 = 6000.0;
        void *outs[2] = {tp(HIU), tp(HIV)};
        if ((rc = xp_interp_levels(&dv[6], 2, vars, 1, &at6, 0, outs, stream))) return rc;
    }
    // 4. per point
    xp::ConvFinishArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.ncol = ncol; fa.ignore_nans = ignore_nans != 0;
    fa.mu_p = tp(MUP); fa.mu_td = tp(MUTD);
    fa.li[0] = o_[O_MU_LI]; fa.li[1] = o_[O_M1_LI]; fa.li[2] = o_[O_M5_LI];
    fa.t850 = tp(T850); fa.t700 = tp(T700); fa.t500 = tp(T500); fa.td850 = tp(TD850); fa.z700 = tp(Z700); fa.z500 = tp(Z500);
    fa.hi_u = tp(HIU); fa.hi_v = tp(HIV); fa.sfc_u = sfu; fa.sfc_v = sfv; fa.valid = (const int32_t *)valid;
    fa.cape[0] = o_[O_MU_CAPE]; fa.cape[1] = o_[O_M1_CAPE]; fa.cape[2] = o_[O_M5_CAPE];
    fa.cin[0] = o_[O_MU_CIN]; fa.cin[1] = o_[O_M1_CIN]; fa.cin[2] = o_[O_M5_CIN];
    fa.li_out[0] = o_[O_MU_LI]; fa.li_out[1] = o_[O_M1_LI]; fa.li_out[2] = o_[O_M5_LI];
    fa.freezing = o_[O_FRZ]; fa.melting = o_[O_MLT];
    fa.mu_mixing_ratio = o_[O_MU_W]; fa.dci[0] = o_[O_MU_DCI]; fa.dci[1] = o_[O_M1_DCI]; fa.dci[2] = o_[O_M5_DCI];
    fa.lapse = o_[O_LAPSE]; fa.temp_500 = o_[O_T500]; fa.shear_u = o_[O_SHU]; fa.shear_v = o_[O_SHV]; fa.shear_mag = o_[O_SHM];
    fa.positive_shear = (int32_t *)o_[O_POS];
    if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_conv_finish<double>), dim3(blocks(ncol)), dim3(256), 0, st.s, fa);
    else hipLaunchKernelGGL((xp::k_conv_finish<float>), dim3(blocks(ncol)), dim3(256), 0, st.s, fa);
    return st.finish();
}

int xp_select_parcel(const xp_view *p, const xp_view *t, const xp_view *td, const xp_parcel *parcel,
                     xp_scalars_out *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    Stager st(stream);
    xp::CapeArgs a;
    if ((rc = fill_common(st, p, t, td, parcel, nullptr, &a))) return rc;
    if (parcel->mode != XP_PARCEL_MOST_UNSTABLE && parcel->mode != XP_PARCEL_MIXED_LAYER)
        return fail(XP_E_ARG, "xp_select_parcel: mode must be most-unstable or mixed-layer");
    if ((rc = stage_scalars(st, out, a.ncol, &a.s))) return rc;
    if (a.ncol) {
        dim3 gr(blocks(a.ncol)), bl(256);
        if (p->dtype == XP_F64) {
            if (parcel->mode == XP_PARCEL_MOST_UNSTABLE) hipLaunchKernelGGL((xp::k_select_parcel<double, xp::PM_MU>), gr, bl, 0, st.s, a);
            else hipLaunchKernelGGL((xp::k_select_parcel<double, xp::PM_ML>), gr, bl, 0, st.s, a);
        } else {
            if (parcel->mode == XP_PARCEL_MOST_UNSTABLE) hipLaunchKernelGGL((xp::k_select_parcel<float, xp::PM_MU>), gr, bl, 0, st.s, a);
            else hipLaunchKernelGGL((xp::k_select_parcel<float, xp::PM_ML>), gr, bl, 0, st.s, a);
        }
    }
    return st.finish();
}

int xp_mixed_layer(const xp_view *p, const xp_view *v, double depth, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure")) || (rc = check_view(v, "variable")) || (rc = same_shape(p, v, "pressure/variable"))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_mixed_layer: null output");
    Stager st(stream);
    xp::View pv, vv;
    void *od;
    if ((rc = stage_view(st, p, &pv)) || (rc = stage_view(st, v, &vv)) ||
        (rc = st.out(out, (size_t)p->ncol * esize(p->dtype), p->mem, &od))) return rc;
    if (p->ncol) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_mixed_layer<double>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, vv, p->nlev, p->ncol, depth, od, 1);
        else hipLaunchKernelGGL((xp::k_mixed_layer<float>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, vv, p->nlev, p->ncol, depth, od, 0);
    }
    return st.finish();
}

int xp_lcl(int64_t n, int32_t dtype, int32_t mem, const void *pp, const void *pt, const void *ptd, void *lp, void *lt,
           void *ltv, int32_t *status, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (n < 0 || !pp || !pt || !ptd) return fail(XP_E_ARG, "xp_lcl: null input");
    if (dtype != XP_F32 && dtype != XP_F64) return fail(XP_E_ARG, "xp_lcl: bad dtype");
    Stager st(stream);
    size_t b = (size_t)n * esize(dtype);
    const void *dp, *dt, *dtd;
    void *op, *ot, *otv, *os;
    if ((rc = st.in(pp, b, mem, &dp)) || (rc = st.in(pt, b, mem, &dt)) || (rc = st.in(ptd, b, mem, &dtd)) ||
        (rc = st.out(lp, b, mem, &op)) || (rc = st.out(lt, b, mem, &ot)) || (rc = st.out(ltv, b, mem, &otv)) ||
        (rc = st.out(status, (size_t)n * 4, mem, &os))) return rc;
    if (n) {
        if (dtype == XP_F64) hipLaunchKernelGGL((xp::k_lcl<double>), dim3(blocks(n)), dim3(256), 0, st.s, n, dp, dt, dtd, op, ot, otv, (int32_t *)os);
        else hipLaunchKernelGGL((xp::k_lcl<float>), dim3(blocks(n)), dim3(256), 0, st.s, n, dp, dt, dtd, op, ot, otv, (int32_t *)os);
    }
    return st.finish();
}

int xp_dry_lapse(const xp_view *p, const void *pt, const void *pp, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure"))) return rc;
    if (!pt || !out) return fail(XP_E_ARG, "xp_dry_lapse: null argument");
    Stager st(stream);
    xp::View pv; xp::OutView ov;
    size_t cb = (size_t)p->ncol * esize(p->dtype), fb = cb * (size_t)p->nlev;
    const void *dt, *dp;
    void *od;
    if ((rc = stage_view(st, p, &pv)) || (rc = st.in(pt, cb, p->mem, &dt)) || (rc = st.in(pp, cb, p->mem, &dp)) ||
        (rc = st.out(out, fb, p->mem, &od))) return rc;
    ov.data = od; ov.ls = p->lev_stride; ov.cs = p->col_stride;
    if (p->ncol) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_dry_lapse<double>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, p->nlev, p->ncol, dt, dp, ov);
        else hipLaunchKernelGGL((xp::k_dry_lapse<float>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, p->nlev, p->ncol, dt, dp, ov);
    }
    return st.finish();
}

int xp_moist_lapse(const xp_view *p, const void *pt, const void *pp, int32_t moist_mode, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure"))) return rc;
    if (!pt || !out) return fail(XP_E_ARG, "xp_moist_lapse: null argument");
    if (moist_mode == XP_MOIST_TABLE && !g.tables) return fail(XP_E_NO_TABLES, "Call load_moist_adiabat_lookups first.");
    Stager st(stream);
    xp::View pv; xp::OutView ov;
    size_t cb = (size_t)p->ncol * esize(p->dtype), fb = cb * (size_t)p->nlev;
    const void *dt, *dp;
    void *od;
    if ((rc = stage_view(st, p, &pv)) || (rc = st.in(pt, cb, p->mem, &dt)) || (rc = st.in(pp, cb, p->mem, &dp)) ||
        (rc = st.out(out, fb, p->mem, &od))) return rc;
    ov.data = od; ov.ls = p->lev_stride; ov.cs = p->col_stride;
    xp::Tables tb = g.tb;
    int tm = moist_mode == XP_MOIST_TABLE;   /* XP_MOIST_FAMILY is served by the RK4 stepper in the component kernels */
    if (p->ncol) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_moist_lapse<double>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, p->nlev, p->ncol, dt, dp, tm, tb, (const double *)g.es_tab, ov);
        else hipLaunchKernelGGL((xp::k_moist_lapse<float>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, p->nlev, p->ncol, dt, dp, tm, tb, (const double *)g.es_tab, ov);
    }
    return st.finish();
}

int xp_parcel_profile(const xp_view *p, const void *pp, const void *pt, const void *ptd, int32_t moist_mode,
                      void *t_out, void *tv_out, void *lp, void *lt, void *ltv, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure"))) return rc;
    if (!pp || !pt || !ptd) return fail(XP_E_ARG, "xp_parcel_profile: null parcel");
    if (moist_mode == XP_MOIST_TABLE && !g.tables) return fail(XP_E_NO_TABLES, "Call load_moist_adiabat_lookups first.");
    Stager st(stream);
    xp::View pv; xp::OutView ot, otv;
    size_t cb = (size_t)p->ncol * esize(p->dtype), fb = cb * (size_t)p->nlev;
    const void *dpp, *dpt, *dptd;
    void *d1, *d2, *d3, *d4, *d5;
    if ((rc = stage_view(st, p, &pv)) || (rc = st.in(pp, cb, p->mem, &dpp)) || (rc = st.in(pt, cb, p->mem, &dpt)) ||
        (rc = st.in(ptd, cb, p->mem, &dptd)) || (rc = st.out(t_out, fb, p->mem, &d1)) || (rc = st.out(tv_out, fb, p->mem, &d2)) ||
        (rc = st.out(lp, cb, p->mem, &d3)) || (rc = st.out(lt, cb, p->mem, &d4)) || (rc = st.out(ltv, cb, p->mem, &d5))) return rc;
    ot.data = d1; ot.ls = p->lev_stride; ot.cs = p->col_stride;
    otv.data = d2; otv.ls = p->lev_stride; otv.cs = p->col_stride;
    xp::Tables tb = g.tb;
    int tm = moist_mode == XP_MOIST_TABLE;   /* XP_MOIST_FAMILY is served by the RK4 stepper in the component kernels */
    if (p->ncol) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_parcel_profile<double>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, p->nlev, p->ncol, dpp, dpt, dptd, tm, tb, (const double *)g.es_tab, ot, otv, d3, d4, d5);
        else hipLaunchKernelGGL((xp::k_parcel_profile<float>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, p->nlev, p->ncol, dpp, dpt, dptd, tm, tb, (const double *)g.es_tab, ot, otv, d3, d4, d5);
    }
    return st.finish();
}

int xp_lfc_el(const xp_view *p, const xp_view *par, const xp_view *env, const void *lcl_p, const void *lcl_t,
              xp_scalars_out *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure")) || (rc = check_view(par, "parcel_temperature")) || (rc = check_view(env, "temperature")) ||
        (rc = same_shape(p, par, "pressure/parcel_temperature")) || (rc = same_shape(p, env, "pressure/temperature"))) return rc;
    if (!lcl_p || !lcl_t || !out) return fail(XP_E_ARG, "xp_lfc_el: null argument");
    Stager st(stream);
    xp::View pv, pav, ev;
    xp::ScalarsOut so;
    size_t cb = (size_t)p->ncol * esize(p->dtype);
    const void *dlp, *dlt;
    if ((rc = stage_view(st, p, &pv)) || (rc = stage_view(st, par, &pav)) || (rc = stage_view(st, env, &ev)) ||
        (rc = st.in(lcl_p, cb, p->mem, &dlp)) || (rc = st.in(lcl_t, cb, p->mem, &dlt)) ||
        (rc = stage_scalars(st, out, p->ncol, &so))) return rc;
    if (p->ncol) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_lfc_el<double>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, pav, ev, p->nlev, p->ncol, dlp, dlt, so);
        else hipLaunchKernelGGL((xp::k_lfc_el<float>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, pav, ev, p->nlev, p->ncol, dlp, dlt, so);
    }
    return st.finish();
}

int xp_cape_cin_base(const xp_view *p, const xp_view *env, const xp_view *par, const void *lfc_p, const void *el_p,
                     const xp_opts *o, void *cape, void *cin, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure")) || (rc = check_view(par, "parcel_temperature")) || (rc = check_view(env, "temperature")) ||
        (rc = same_shape(p, par, "pressure/parcel_temperature")) || (rc = same_shape(p, env, "pressure/temperature"))) return rc;
    if (!lfc_p || !el_p) return fail(XP_E_ARG, "xp_cape_cin_base: null argument");
    Stager st(stream);
    xp::View pv, pav, ev;
    size_t cb = (size_t)p->ncol * esize(p->dtype);
    const void *dl, *de;
    void *dc, *dn;
    if ((rc = stage_view(st, p, &pv)) || (rc = stage_view(st, par, &pav)) || (rc = stage_view(st, env, &ev)) ||
        (rc = st.in(lfc_p, cb, p->mem, &dl)) || (rc = st.in(el_p, cb, p->mem, &de)) || (rc = st.out(cape, cb, p->mem, &dc)) ||
        (rc = st.out(cin, cb, p->mem, &dn))) return rc;
    int pn = o ? o->pos_cape_neg_cin : 1, pz = o ? o->post_zero_cin : 0;
    if (p->ncol) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_cape_cin_base<double>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, ev, pav, p->nlev, p->ncol, dl, de, pn, pz, dc, dn);
        else hipLaunchKernelGGL((xp::k_cape_cin_base<float>), dim3(blocks(p->ncol)), dim3(256), 0, st.s, pv, ev, pav, p->nlev, p->ncol, dl, de, pn, pz, dc, dn);
    }
    return st.finish();
}

int xp_wet_bulb_temperature(const xp_view *p, const xp_view *t, const xp_view *td, int32_t moist_mode, void *out,
                            void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure")) || (rc = check_view(t, "temperature")) || (rc = check_view(td, "dewpoint")) ||
        (rc = same_shape(p, t, "pressure/temperature")) || (rc = same_shape(p, td, "pressure/dewpoint"))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_wet_bulb_temperature: null output");
    if (moist_mode == XP_MOIST_TABLE && !g.tables) return fail(XP_E_NO_TABLES, "Call load_moist_adiabat_lookups first.");
    Stager st(stream);
    xp::View pv, tv, tdv; xp::OutView ov;
    void *od;
    if ((rc = stage_view(st, p, &pv)) || (rc = stage_view(st, t, &tv)) || (rc = stage_view(st, td, &tdv)) ||
        (rc = st.out(out, (size_t)p->nlev * (size_t)p->ncol * esize(p->dtype), p->mem, &od))) return rc;
    ov.data = od; ov.ls = p->lev_stride; ov.cs = p->col_stride;
    xp::Tables tb = g.tb;
    int tm = moist_mode == XP_MOIST_TABLE;   /* XP_MOIST_FAMILY is served by the RK4 stepper in the component kernels */
    int64_t n = p->nlev * p->ncol;
    if (n) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_wet_bulb<double>), dim3(blocks(n)), dim3(256), 0, st.s, pv, tv, tdv, p->nlev, p->ncol, tm, tb, (const double *)g.es_tab, ov);
        else hipLaunchKernelGGL((xp::k_wet_bulb<float>), dim3(blocks(n)), dim3(256), 0, st.s, pv, tv, tdv, p->nlev, p->ncol, tm, tb, (const double *)g.es_tab, ov);
    }
    return st.finish();
}

int xp_interp_level(const xp_view *coords, const xp_view *x, const void *at, int32_t at_is_scalar, int32_t log_coords,
                    void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(coords, "coords")) || (rc = check_view(x, "variable")) || (rc = same_shape(coords, x, "coords/variable"))) return rc;
    if (!at || !out) return fail(XP_E_ARG, "xp_interp_level: null argument");
    Stager st(stream);
    xp::View cv, xv;
    const void *da;
    void *od;
    size_t cb = (size_t)coords->ncol * esize(coords->dtype);
    if ((rc = stage_view(st, coords, &cv)) || (rc = stage_view(st, x, &xv)) ||
        (rc = st.in(at, at_is_scalar ? esize(coords->dtype) : cb, coords->mem, &da)) ||
        (rc = st.out(out, cb, coords->mem, &od))) return rc;
    if (coords->ncol) {
        if (coords->dtype == XP_F64) hipLaunchKernelGGL((xp::k_interp_level<double>), dim3(blocks(coords->ncol)), dim3(256), 0, st.s, cv, xv, coords->nlev, coords->ncol, da, (int)at_is_scalar, (int)log_coords, od);
        else hipLaunchKernelGGL((xp::k_interp_level<float>), dim3(blocks(coords->ncol)), dim3(256), 0, st.s, cv, xv, coords->nlev, coords->ncol, da, (int)at_is_scalar, (int)log_coords, od);
    }
    return st.finish();
}

int xp_interp_levels(const xp_view *coords, int32_t nvar, const xp_view *const *variables, int32_t ntarget, const double *at,
                     int32_t log_coords, void *const *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if (nvar < 1 || nvar > 4 || ntarget < 1 || ntarget > 4) return fail(XP_E_ARG, "xp_interp_levels: 1..4 variables and 1..4 coordinates");
    if (!variables || !at || !out) return fail(XP_E_ARG, "xp_interp_levels: null argument");
    if ((rc = check_view(coords, "coords"))) return rc;
    Stager st(stream);
    xp::View cv;
    xp::InterpMany m;
    memset(&m, 0, sizeof(m));
    if ((rc = stage_view(st, coords, &cv))) return rc;
    size_t cb = (size_t)coords->ncol * esize(coords->dtype);
    for (int v = 0; v < nvar; ++v) {
        if ((rc = check_view(variables[v], "variable")) || (rc = same_shape(coords, variables[v], "coords/variable")) ||
            (rc = stage_view(st, variables[v], &m.x[v]))) return rc;
        for (int j = 0; j < ntarget; ++j)
            if ((rc = st.out(out[v * ntarget + j], cb, coords->mem, &m.out[v * ntarget + j]))) return rc;
    }
    for (int j = 0; j < ntarget; ++j) m.at[j] = at[j];
    if (coords->ncol) {
        if (coords->dtype == XP_F64) launch_interp_levels_v<double>(nvar, ntarget, cv, m, coords->nlev, coords->ncol, (int)log_coords, st.s);
        else launch_interp_levels_v<float>(nvar, ntarget, cv, m, coords->nlev, coords->ncol, (int)log_coords, st.s);
    }
    return st.finish();
}

int xp_dewpoint_from_specific_humidity(const xp_view *p, const xp_view *t, const xp_view *q, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(p, "pressure")) || (rc = check_view(t, "temperature")) || (rc = check_view(q, "specific_humidity")) ||
        (rc = same_shape(p, t, "pressure/temperature")) || (rc = same_shape(p, q, "pressure/specific_humidity"))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_dewpoint_from_specific_humidity: null output");
    Stager st(stream);
    xp::View pv, tv, qv; xp::OutView ov;
    void *od;
    if ((rc = stage_view(st, p, &pv)) || (rc = stage_view(st, t, &tv)) || (rc = stage_view(st, q, &qv)) ||
        (rc = st.out(out, (size_t)p->nlev * (size_t)p->ncol * esize(p->dtype), p->mem, &od))) return rc;
    ov.data = od; ov.ls = p->lev_stride; ov.cs = p->col_stride;
    int64_t n = p->nlev * p->ncol;
    if (n) {
        if (p->dtype == XP_F64) hipLaunchKernelGGL((xp::k_dewpoint_from_q<double>), dim3(blocks(n)), dim3(256), 0, st.s, pv, tv, qv, p->nlev, p->ncol, ov);
        else hipLaunchKernelGGL((xp::k_dewpoint_from_q<float>), dim3(blocks(n)), dim3(256), 0, st.s, pv, tv, qv, p->nlev, p->ncol, ov);
    }
    return st.finish();
}

int xp_crossing_level(const xp_view *x, const xp_view *a, double value, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(x, "x")) || (rc = check_view(a, "a")) || (rc = same_shape(x, a, "x/a"))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_crossing_level: null output");
    Stager st(stream);
    xp::View xv, av;
    void *od;
    if ((rc = stage_view(st, x, &xv)) || (rc = stage_view(st, a, &av)) ||
        (rc = st.out(out, (size_t)x->ncol * esize(x->dtype), x->mem, &od))) return rc;
    if (x->ncol) {
        if (x->dtype == XP_F64) hipLaunchKernelGGL((xp::k_crossing_level<double>), dim3(blocks(x->ncol)), dim3(256), 0, st.s, xv, av, x->nlev, x->ncol, value, od);
        else hipLaunchKernelGGL((xp::k_crossing_level<float>), dim3(blocks(x->ncol)), dim3(256), 0, st.s, xv, av, x->nlev, x->ncol, value, od);
    }
    return st.finish();
}

int xp_mixing_ratio(const xp_view *t, const xp_view *td, const xp_view *p, void *out, void *stream) {
    DevGuard dg_;
    int rc = ensure_init();
    if (rc) return rc;
    if ((rc = check_view(t, "temperature")) || (rc = check_view(td, "dewpoint")) || (rc = check_view(p, "pressure")) ||
        (rc = same_shape(t, td, "temperature/dewpoint")) || (rc = same_shape(t, p, "temperature/pressure"))) return rc;
    if (!out) return fail(XP_E_ARG, "xp_mixing_ratio: null output");
    Stager st(stream);
    xp::View tv, tdv, pv; xp::OutView ov;
    void *od;
    if ((rc = stage_view(st, t, &tv)) || (rc = stage_view(st, td, &tdv)) || (rc = stage_view(st, p, &pv)) ||
        (rc = st.out(out, (size_t)t->nlev * (size_t)t->ncol * esize(t->dtype), t->mem, &od))) return rc;
    ov.data = od; ov.ls = t->lev_stride; ov.cs = t->col_stride;
    int64_t n = t->nlev * t->ncol;
    if (n) {
        if (t->dtype == XP_F64) hipLaunchKernelGGL((xp::k_mixing_ratio<double>), dim3(blocks(n)), dim3(256), 0, st.s, tv, tdv, pv, t->nlev, t->ncol, ov);
        else hipLaunchKernelGGL((xp::k_mixing_ratio<float>), dim3(blocks(n)), dim3(256), 0, st.s, tv, tdv, pv, t->nlev, t->ncol, ov);
    }
    return st.finish();
}

}  // extern "C"

// the reference's array primitives (insert_level, find_intersections, trapz, ...): kernels + entry points
#include "xp_primitives_abi.hpp"
