/*
 * oracle/c/xp_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, plain C99 + OpenMP).
 *
 * One-column-at-a-time restatement of the reference's parcel-lifting path
 * (traupach/xarray_parcel, modules/parcel_functions.py = "pf.py"), written the way
 * the reference computes: the LCL is physically inserted as an extra level
 * (N -> N+1 arrays), intersections are materialised as arrays, areas are built as
 * arrays and reduced with skip-NaN sums.  It is deliberately NOT organised like the
 * HIP kernel (which streams each column once with the LCL as a virtual level), so
 * that agreement between the two means something.
 *
 * Thermodynamics: MetPy 1.4.1 formulas (un-vendored dependency of the reference,
 * pinned by parcel_functions_demo.ipynb:86), see oracle/thermo.py for citations.
 * This file is validated against oracle/parcel_oracle.py (which is pinned by all
 * the reference's KATs) in tests/test_c_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file.  The product never links it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RD 287.04749097718457
#define EPSILON 0.6219569100577033
#define KAPPA (2.0 / 7.0)
#define CP_D (RD / KAPPA)
#define LV 2.50084e6
#define VT_EPS 0.608
#define RK4_H_MAX 0.1
#define FILL (-999.0)

typedef struct {
    int32_t vtc;               /* virtual_temperature_correction (pf.py:1396) */
    int32_t lcl_interp_log;    /* lcl_interp == 'log' (pf.py:1396)            */
    int32_t pos_cape_neg_cin;  /* pf.py:1293 */
    int32_t post_zero_cin;     /* pf.py:1293 */
    int32_t parcel_mode;       /* 0 surface, 1 most unstable, 2 mixed layer, 3 explicit */
    int32_t moist_mode;        /* 0 rk4 spec, 1 reference lookup tables, 2 adiabat family (oracle/family.py) */
    double depth;              /* hPa; MU default 300 (pf.py:1558), ML default 100 (pf.py:1652) */
} xpo_opts;

typedef struct {               /* reference-table mode (pf.py:447-523), see oracle/tables.py */
    int64_t n_p, n_t, n_adiabat;
    double p_max, p_step, t_min, t_step;  /* index grid: p = p_max - i*p_step, t = t_min + j*t_step */
    const uint16_t *index;     /* [n_p][n_t], 0 = NaN */
    const float *adiabats;     /* [n_adiabat][n_p], pressure ASCENDING (pf.py:54) */
} xpo_tables;

static xpo_tables g_tables;
static int g_tables_loaded = 0;

void xpo_set_tables(const xpo_tables *t) { g_tables = *t; g_tables_loaded = 1; }

/* ---- MetPy 1.4.1 thermo --------------------------------------------------------- */
static double es(double t) { return 6.112 * exp(17.67 * (t - 273.15) / (t - 29.65)); }
static double dewpoint_of_e(double e) { double v = log(e / 6.112); return 273.15 + 243.5 * v / (17.67 - v); }
static double mix_of_e(double e, double p) { return EPSILON * e / (p - e); }
static double sat_mix(double p, double t) { return mix_of_e(es(t), p); }
static double vapor_pressure(double p, double w) { return p * w / (EPSILON + w); }
/* pf.py:684-710: RH(T,Td) * w_s(p,T) */
static double mixing_ratio(double t, double td, double p) { return (es(td) / es(t)) * sat_mix(p, t); }
static double virt(double t, double w) { return t * (1 + VT_EPS * w); }
static double theta_e(double p, double t, double td) {
    double e = es(td), r = sat_mix(p, td);
    double tl = 56.0 + 1.0 / (1.0 / (td - 56.0) + log(t / td) / 800.0);
    double thl = t / pow((p - e) / 1000.0, KAPPA) * pow(t / tl, 0.28 * r);
    return thl * exp(r * (1.0 + 0.448 * r) * (3036.0 / tl - 1.78));
}

/* skip-NaN reductions with xarray semantics */
static double nmax(const double *x, int n) { double m = NAN; for (int i = 0; i < n; i++) if (!isnan(x[i]) && !(x[i] <= m)) m = x[i]; return m; }
static double nmin(const double *x, int n) { double m = NAN; for (int i = 0; i < n; i++) if (!isnan(x[i]) && !(x[i] >= m)) m = x[i]; return m; }

/* ---- LCL: per-column Steffensen spelling of metpy.calc.lcl (oracle/thermo.py lcl_steffensen) */
static double lcl_iter(double p, double p0, double w, double t) {
    double td = dewpoint_of_e(vapor_pressure(p, w));
    return p0 * pow(td / t, 1.0 / KAPPA);
}
static int np_isclose(double a, double b) { return fabs(a - b) <= 1e-8 + 1e-5 * fabs(b); }

/* returns 0 ok, 1 not converged */
int xpo_lcl(double p_start, double t, double td, double *p_lcl, double *t_lcl, double *tv_lcl) {
    if (isnan(p_start) || isnan(t) || isnan(td)) { *p_lcl = *t_lcl = *tv_lcl = NAN; return 0; }  /* pf.py:627-634,680 */
    double w = mix_of_e(es(td), p_start);
    double p0 = p_start, p = NAN;
    int conv = 0;
    for (int it = 0; it < 50; it++) {
        double p1 = lcl_iter(p0, p_start, w, t);
        double p2 = lcl_iter(p1, p_start, w, t);
        double d = p2 - 2.0 * p1 + p0;
        p = (d != 0) ? p0 - (p1 - p0) * (p1 - p0) / d : p2;
        double rel = (p0 != 0) ? (p - p0) / p0 : p;
        if (fabs(rel) < 1e-5) { conv = 1; break; }
        p0 = p;
    }
    if (!conv) p = NAN;
    if (np_isclose(p, p_start)) p = p_start;
    *p_lcl = p;
    *t_lcl = dewpoint_of_e(vapor_pressure(p, w));
    *tv_lcl = virt(*t_lcl, mixing_ratio(*t_lcl, *t_lcl, p));    /* pf.py:653-657 */
    return conv ? 0 : 1;
}

/* ---- moist adiabat ---------------------------------------------------------------- */
static double dt_dlnp(double x, double t) {
    double p = exp(x), e = es(t), pe = p - e;
    double num = RD * t * pe + LV * EPSILON * e;
    double den = CP_D * RD * t * t * pe + LV * LV * EPSILON * EPSILON * e;
    return RD * t * t * num / den;
}
static int rk4_substeps(double dx) { int n = (int)ceil(fabs(dx) / RK4_H_MAX - 1e-12); return n < 1 ? 1 : n; }

/* RK4 spec (oracle/thermo.py moist_lapse_rk4): march away from the reference point on each side */
static void moist_lapse_rk4(int n, const double *p, double t0, double pref, double *out) {
    int *ord = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) out[i] = NAN;
    if (isnan(pref) || isnan(t0)) { free(ord); return; }
    double xref = log(pref);
    for (int side = 0; side < 2; side++) {
        int m = 0;
        for (int i = 0; i < n; i++) {
            if (isnan(p[i])) continue;
            if ((side == 0 && p[i] <= pref) || (side == 1 && p[i] > pref)) ord[m++] = i;
        }
        for (int i = 1; i < m; i++) {           /* stable insertion sort by |ln p - ln pref| */
            int k = ord[i]; double dk = fabs(log(p[k]) - xref); int j = i - 1;
            while (j >= 0 && fabs(log(p[ord[j]]) - xref) > dk) { ord[j + 1] = ord[j]; j--; }
            ord[j + 1] = k;
        }
        double x = xref, t = t0;
        for (int q = 0; q < m; q++) {
            int k = ord[q];
            double x1 = log(p[k]);
            if (x1 != x) {
                int ns = rk4_substeps(x1 - x);
                double h = (x1 - x) / ns, xs = x;
                for (int s = 0; s < ns; s++) {
                    double k1 = dt_dlnp(xs, t);
                    double k2 = dt_dlnp(xs + 0.5 * h, t + 0.5 * h * k1);
                    double k3 = dt_dlnp(xs + 0.5 * h, t + 0.5 * h * k2);
                    double k4 = dt_dlnp(xs + h, t + h * k3);
                    t = t + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
                    xs = xs + h;
                }
                x = x1;
            }
            out[k] = t;
        }
    }
    free(ord);
}

/* reference lookup-table mode, pf.py:525-607 (semantics: SURVEY A.4, oracle/tables.py) */
static void moist_lapse_table(int n, const double *p, double t0, double pref, double *out) {
    const xpo_tables *T = &g_tables;
    for (int i = 0; i < n; i++) out[i] = NAN;
    if (!g_tables_loaded || isnan(t0) || isnan(pref)) return;
    /* nearest neighbour on both coordinates with pandas' tie rules (oracle/tables.py nearest_index_*): on the
       DEcreasing pressure index the left (higher-pressure) label wins a tie, on the increasing temperature index
       the right one */
    double fi = (T->p_max - pref) / T->p_step, fj = (t0 - T->t_min) / T->t_step;
    double i0 = floor(fi), j0 = floor(fj);
    int64_t ip = (int64_t)(((i0 + 1.0) - fi < fi - i0) ? i0 + 1.0 : i0);
    int64_t jt = (int64_t)(((j0 + 1.0) - fj <= fj - j0) ? j0 + 1.0 : j0);
    if (ip < 0) ip = 0;
    if (ip > T->n_p - 1) ip = T->n_p - 1;
    if (jt < 0) jt = 0;
    if (jt > T->n_t - 1) jt = T->n_t - 1;
    uint16_t a = T->index[ip * T->n_t + jt];
    if (a == 0) return;                                          /* NaN cell: pf.py:570-582 */
    const float *row = T->adiabats + (int64_t)(a - 1) * T->n_p; /* ascending p: row[k] at p_min + k*step */
    double p_min = T->p_max - (double)(T->n_p - 1) * T->p_step;
    for (int k = 0; k < n; k++) {
        double pk = p[k];
        if (isnan(pk) || pk < p_min || pk > T->p_max) continue;  /* pf.py:598-605 */
        double f = (pk - p_min) / T->p_step;
        int64_t lo = (int64_t)floor(f);
        if (lo >= T->n_p - 1) { out[k] = row[T->n_p - 1]; continue; }
        double xlo = p_min + (double)lo * T->p_step, xhi = p_min + (double)(lo + 1) * T->p_step;
        double ylo = row[lo], yhi = row[lo + 1];
        /* np.interp: slope * (x - xlo) + ylo */
        out[k] = (yhi - ylo) / (xhi - xlo) * (pk - xlo) + ylo;
    }
}

/* ---- "adiabat family" exact mode (specification: oracle/family.py) ------------------------------------------- */
/* Tv(x ; psi) = sum_n sum_m A[j][n][m][q] z^n s^m on x-pieces j (width 0.5 in ln p below ln 1100) and psi-pieces q
 * (EDGES): the parcel's VIRTUAL temperature T (1 + 0.608 w_s(p, T)) along the pseudo-adiabat with T = psi at 1000 hPa;
 * A = monomial form of the 9 x 9 Chebyshev-node interpolant; layout [j][n][m][q].  The parcel temperature is the T
 * that has this virtual temperature (Newton, fam_temperature_of). */
#define FAM_XHI 7.003065458786462 /* ln 1100 */
#define FAM_WX 0.5
#define FAM_NPX 8
#define FAM_XLO (FAM_XHI - FAM_WX * FAM_NPX)
#define FAM_ND 8
#define FAM_MD 8
#define FAM_NPS 9
#define FAM_X1000 6.907755278982137
#define FAM_BUILD_H 0.0125
#define FAM_LABEL_H 0.25
#define FAM_MARGIN 0.05
#define FAM_NEWTON 3
#define FAM_SIZE (FAM_NPX * (FAM_ND + 1) * (FAM_MD + 1) * FAM_NPS)
static const double FAM_EDGES[FAM_NPS + 1] = {215.0, 245.0, 262.0, 275.0, 285.0, 293.0, 299.0, 304.0, 308.5, 312.0};
static double *g_fam = NULL;

static double fam_rk4(double x, double t, double x1, double h_max) {
    int n = (int)ceil(fabs(x1 - x) / h_max - 1e-12);
    if (n < 1) n = 1;
    double h = (x1 - x) / n;
    for (int s = 0; s < n; s++) {
        double k1 = dt_dlnp(x, t);
        double k2 = dt_dlnp(x + 0.5 * h, t + 0.5 * h * k1);
        double k3 = dt_dlnp(x + 0.5 * h, t + 0.5 * h * k2);
        double k4 = dt_dlnp(x + h, t + h * k3);
        t = t + h / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4);
        x = x + h;
    }
    return t;
}
static double fam_virtual_temperature(double p, double t) {
    double e = es(t);
    return t * (1.0 + VT_EPS * (EPSILON * e / (p - e)));
}
/* the T with fam_virtual_temperature(p, T) = tv: five Newton steps from T0 = tv / (1 + 0.608 w_s(p, tv)) */
static double fam_temperature_of(double p, double tv) {
    const double c = VT_EPS * EPSILON;
    double e0 = es(tv);
    double t = tv / (1.0 + VT_EPS * (EPSILON * e0 / (p - e0)));
    for (int it = 0; it < 5; it++) {
        double e = es(t);
        double de = e * (17.67 * 243.5) / ((t - 29.65) * (t - 29.65));
        double f = t * (1.0 + c * e / (p - e)) - tv;
        double df = 1.0 + c * e / (p - e) + t * c * p * de / ((p - e) * (p - e));
        t = t - f / df;
    }
    return t;
}
static double fam_xmid(int j) { return FAM_XHI - FAM_WX * (j + 0.5); }
static double fam_smid(int q) { return 0.5 * (FAM_EDGES[q] + FAM_EDGES[q + 1]); }
static double fam_shalf(int q) { return 0.5 * (FAM_EDGES[q + 1] - FAM_EDGES[q]); }
/* solve V c = y for the monomial coefficients of the interpolant through (u_k, y_k), k < n (n <= 9) */
static void fam_monomials(int n, const double *u, const double *y, double *c) {
    long double A[9][10];
    for (int k = 0; k < n; k++) {
        long double v = 1.0L;
        for (int i = 0; i < n; i++) { A[k][i] = v; v *= (long double)u[k]; }
        A[k][n] = (long double)y[k];
    }
    for (int col = 0; col < n; col++) {
        int piv = col;
        for (int r = col + 1; r < n; r++) if (fabsl(A[r][col]) > fabsl(A[piv][col])) piv = r;
        for (int i = 0; i <= n; i++) { long double t_ = A[col][i]; A[col][i] = A[piv][i]; A[piv][i] = t_; }
        for (int r = 0; r < n; r++) {
            if (r == col) continue;
            long double f = A[r][col] / A[col][col];
            for (int i = col; i <= n; i++) A[r][i] -= f * A[col][i];
        }
    }
    for (int i = 0; i < n; i++) c[i] = (double)(A[i][n] / A[i][i]);
}
static void fam_build(double *tab) {
    enum { NN = FAM_ND + 1, MM = FAM_MD + 1, NXN = FAM_NPX * (FAM_ND + 1), NSN = FAM_NPS * (FAM_MD + 1) };
    const double pi = 3.14159265358979323846;
    double un[NN], um[MM], xs[NXN], ps[NSN];
    double *vals = (double *)malloc(sizeof(double) * NXN * NSN);
    for (int k = 0; k < NN; k++) un[k] = cos(pi * (k + 0.5) / NN);
    for (int k = 0; k < MM; k++) um[k] = cos(pi * (k + 0.5) / MM);
    for (int j = 0; j < FAM_NPX; j++) for (int k = 0; k < NN; k++) xs[j * NN + k] = fam_xmid(j) + 0.5 * FAM_WX * un[k];
    for (int q = 0; q < FAM_NPS; q++) for (int k = 0; k < MM; k++) ps[q * MM + k] = fam_smid(q) + fam_shalf(q) * um[k];
    /* x-nodes in order of distance from ln 1000, each side separately */
    for (int side = 0; side < 2; side++) {
        int order[NXN], cnt = 0;
        for (int i = 0; i < NXN; i++) if ((side == 0) == (xs[i] <= FAM_X1000)) order[cnt++] = i;
        for (int a = 1; a < cnt; a++) {                    /* insertion sort by |x - ln 1000| (stable) */
            int v = order[a], b = a - 1;
            while (b >= 0 && fabs(xs[order[b]] - FAM_X1000) > fabs(xs[v] - FAM_X1000)) { order[b + 1] = order[b]; b--; }
            order[b + 1] = v;
        }
        for (int c = 0; c < NSN; c++) {
            double x = FAM_X1000, t = ps[c];
            for (int a = 0; a < cnt; a++) {
                int i = order[a];
                if (xs[i] != x) { t = fam_rk4(x, t, xs[i], FAM_BUILD_H); x = xs[i]; }
                vals[(size_t)i * NSN + c] = fam_virtual_temperature(exp(xs[i]), t);
            }
        }
    }
    for (int j = 0; j < FAM_NPX; j++)
        for (int q = 0; q < FAM_NPS; q++) {
            double a[NN][MM], col[NN], cf[NN], row[MM], rf[MM];
            for (int m = 0; m < MM; m++) {                 /* monomials in z for every psi-node ... */
                for (int k = 0; k < NN; k++) col[k] = vals[(size_t)(j * NN + k) * NSN + (q * MM + m)];
                fam_monomials(NN, un, col, cf);
                for (int n = 0; n < NN; n++) a[n][m] = cf[n];
            }
            for (int n = 0; n < NN; n++) {                 /* ... then in s for every z-coefficient */
                for (int m = 0; m < MM; m++) row[m] = a[n][m];
                fam_monomials(MM, um, row, rf);
                for (int m = 0; m < MM; m++) tab[(((size_t)j * NN + n) * MM + m) * FAM_NPS + q] = rf[m];
            }
        }
    free(vals);
}
const double *xpo_family_table(void) {
    /* built on first use, possibly from inside an OpenMP region: publish the pointer only when the table is complete */
    if (!__atomic_load_n(&g_fam, __ATOMIC_ACQUIRE)) {
#pragma omp critical(xpo_fam_init)
        {
            if (!g_fam) {
                double *tab = (double *)malloc(sizeof(double) * FAM_SIZE);
                fam_build(tab);
                __atomic_store_n(&g_fam, tab, __ATOMIC_RELEASE);
            }
        }
    }
    return g_fam;
}
/* tests hand one table to both sides: replace the oracle's own */
void xpo_set_family_table(const double *tab) {
    xpo_family_table();
    memcpy(g_fam, tab, sizeof(double) * FAM_SIZE);
}
static double fam_a(const double *tab, int j, int n, int m, int q) {
    return tab[(((size_t)j * (FAM_ND + 1) + n) * (FAM_MD + 1) + m) * FAM_NPS + q];
}
static int fam_xpiece(double x) {
    double j = floor((FAM_XHI - x) * (1.0 / FAM_WX));
    return j < 0 ? 0 : (j > FAM_NPX - 1 ? FAM_NPX - 1 : (int)j);
}
static int fam_spiece(double psi) {
    int q = 0;
    while (q < FAM_NPS - 1 && psi >= FAM_EDGES[q + 1]) q++;
    return q;
}
/* c_n = sum_m A[j][n][m][q] s^m */
static void fam_column_poly(const double *tab, int j, int q, double s, double *c) {
    for (int n = 0; n <= FAM_ND; n++) {
        double v = fam_a(tab, j, n, FAM_MD, q);
        for (int m = FAM_MD - 1; m >= 0; m--) v = v * s + fam_a(tab, j, n, m, q);
        c[n] = v;
    }
}
static double fam_horner(int deg, const double *c, double u) {
    double v = c[deg];
    for (int k = deg - 1; k >= 0; k--) v = v * u + c[k];
    return v;
}
/* label psi (and its piece q) of the adiabat through (x_lcl, t_lcl); NaN when outside the table */
static double fam_label(const double *tab, double p_lcl, double x_lcl, double t_lcl, int *q_out) {
    *q_out = -1;
    if (!(isfinite(x_lcl) && isfinite(t_lcl)) || !(x_lcl >= FAM_XLO && x_lcl <= FAM_XHI)) return NAN;
    double psi0 = (x_lcl != FAM_X1000) ? fam_rk4(x_lcl, t_lcl, FAM_X1000, FAM_LABEL_H) : t_lcl;
    if (!(psi0 >= FAM_EDGES[0] + FAM_MARGIN && psi0 <= FAM_EDGES[FAM_NPS] - FAM_MARGIN)) return NAN;
    int q = fam_spiece(psi0), j = fam_xpiece(x_lcl);
    const double tv_lcl = fam_virtual_temperature(p_lcl, t_lcl);
    double z = (x_lcl - fam_xmid(j)) * (2.0 / FAM_WX);
    double b[FAM_MD + 1], db[FAM_MD];
    for (int m = 0; m <= FAM_MD; m++) {
        double v = fam_a(tab, j, FAM_ND, m, q);
        for (int n = FAM_ND - 1; n >= 0; n--) v = v * z + fam_a(tab, j, n, m, q);
        b[m] = v;
    }
    for (int m = 1; m <= FAM_MD; m++) db[m - 1] = b[m] * m;
    double inv_h = 1.0 / fam_shalf(q), psi = psi0;
    for (int it = 0; it < FAM_NEWTON; it++) {
        double s = (psi - fam_smid(q)) * inv_h;
        psi = psi - (fam_horner(FAM_MD, b, s) - tv_lcl) / (fam_horner(FAM_MD - 1, db, s) * inv_h);
    }
    if (!(fabs(psi - psi0) <= FAM_MARGIN)) return NAN;
    *q_out = q;
    return psi;
}
static double fam_eval(const double *tab, double x, double psi, int q) {
    if (!isfinite(x) || x > FAM_XHI) return NAN;
    double s = (psi - fam_smid(q)) * (1.0 / fam_shalf(q));
    double c[FAM_ND + 1];
    if (x < FAM_XLO) {                                     /* above the table top: dry continuation */
        fam_column_poly(tab, FAM_NPX - 1, q, s, c);
        return fam_horner(FAM_ND, c, -1.0) * exp(KAPPA * (x - FAM_XLO));
    }
    int j = fam_xpiece(x);
    fam_column_poly(tab, j, q, s, c);
    return fam_horner(FAM_ND, c, (x - fam_xmid(j)) * (2.0 / FAM_WX));
}
static void moist_lapse_family(int n, const double *p, double t0, double pref, double *out) {
    const double *tab = xpo_family_table();
    int q = -1;
    double psi = (pref > 0) ? fam_label(tab, pref, log(pref), t0, &q) : NAN;
    int ok = !isnan(psi);
    for (int k = 0; k < n && ok; k++) {
        if (isnan(p[k])) { out[k] = NAN; continue; }
        out[k] = (p[k] == pref) ? t0 : fam_temperature_of(p[k], fam_eval(tab, log(p[k]), psi, q));
        if (isnan(out[k])) ok = 0;
    }
    if (!ok) moist_lapse_rk4(n, p, t0, pref, out);      /* label or a level outside the table: whole parcel by RK4 */
}

void xpo_moist_lapse(int n, const double *p, double t0, double pref, int moist_mode, double *out) {
    if (moist_mode == 1) moist_lapse_table(n, p, t0, pref, out);
    else if (moist_mode == 2) moist_lapse_family(n, p, t0, pref, out);
    else moist_lapse_rk4(n, p, t0, pref, out);
}

/* ---- L1 array primitives -------------------------------------------------------------- */
/* pf.py:933-990.  d[0] is the coordinate variable; nvar arrays of n -> nvar arrays of n+1. */
static void insert_level(int n, int nvar, double *const *d, const double *level, double **out) {
    double lev_c = level[0];
    for (int v = 0; v < nvar; v++) for (int i = 0; i <= n; i++) out[v][i] = NAN;
    /* pass 1: coordinate with NaN -> FILL, below / above-shifted merge */
    double *cb = (double *)malloc(sizeof(double) * (size_t)(n + 1));   /* coord of 'below' family */
    for (int i = 0; i <= n; i++) cb[i] = NAN;
    for (int i = 0; i < n; i++) {
        double c = isnan(d[0][i]) ? FILL : d[0][i];
        if (c >= lev_c) cb[i] = c;
    }
    for (int v = 0; v < nvar; v++) {
        for (int i = 0; i <= n; i++) {
            double below = NAN, above = NAN;
            if (i < n) {
                int nanrow = isnan(d[0][i]);
                double c = nanrow ? FILL : d[0][i];
                double val = nanrow ? FILL : d[v][i];
                if (c >= lev_c) below = val;
            }
            if (i >= 1) {
                int nanrow = isnan(d[0][i - 1]);
                double c = nanrow ? FILL : d[0][i - 1];
                double val = nanrow ? FILL : d[v][i - 1];
                if (c < lev_c) above = val;
            }
            out[v][i] = isnan(cb[i]) ? above : below;               /* pf.py:977 */
        }
    }
    for (int i = 0; i <= n; i++) cb[i] = out[0][i];                 /* merged coordinate */
    for (int v = 0; v < nvar; v++)
        for (int i = 0; i <= n; i++) {
            if (isnan(cb[i])) out[v][i] = level[v];                 /* pf.py:985 */
            if (out[v][i] == FILL) out[v][i] = NAN;                 /* pf.py:988 */
        }
    free(cb);
}

/* pf.py:1758-1811 on one variable; coords may already be log-transformed */
static void interp_brackets(int n, const double *coords, double at, double *cb, double *ca) {
    double b = NAN, a = NAN;
    for (int i = 0; i < n; i++) {
        double c = coords[i];
        if (isnan(c)) continue;
        if (c >= at && !(c >= b)) b = c;     /* min of coords >= at */
        if (c <= at && !(c <= a)) a = c;     /* max of coords <= at */
    }
    *cb = b; *ca = a;
}
static double mean_where_eq(int n, const double *coords, const double *x, double c) {
    double s = 0; int m = 0;
    for (int i = 0; i < n; i++) if (coords[i] == c && !isnan(x[i])) { s += x[i]; m++; }
    return m ? s / m : NAN;
}
static double linear_interp(int n, const double *coords, const double *x, double at) {
    double cb, ca; interp_brackets(n, coords, at, &cb, &ca);
    double xb = mean_where_eq(n, coords, x, cb), xa = mean_where_eq(n, coords, x, ca);
    double res = xb + (xa - xb) * ((at - cb) / (ca - cb));
    return (xb == xa) ? xb : res;
}

/* pf.py:992-1064.  Arrays of n-1: interval i between levels i and i+1.  cls: +1 increasing,
   -1 decreasing, 0 neither (NaN sign or zero). */
static void find_intersections(int n, const double *x_in, const double *a, const double *b, int log_x,
                               double *ix, double *iy, int *cls) {
    for (int i = 0; i + 1 < n; i++) {
        double s0 = a[i] - b[i], s1 = a[i + 1] - b[i + 1];
        double g0 = (s0 > 0) - (s0 < 0), g1 = (s1 > 0) - (s1 < 0);
        if (isnan(s0)) g0 = NAN;
        if (isnan(s1)) g1 = NAN;
        double diff = g1 - g0;
        ix[i] = NAN; iy[i] = NAN; cls[i] = 0;
        if (diff == 0) continue;                           /* NaN falls through (pf.py:1022) */
        double x0 = log_x ? log(x_in[i]) : x_in[i], x1 = log_x ? log(x_in[i + 1]) : x_in[i + 1];
        double xs = (s1 * x0 - s0 * x1) / (s1 - s0);
        double ys = ((xs - x0) / (x1 - x0)) * (a[i + 1] - a[i]) + a[i];
        ix[i] = log_x ? exp(xs) : xs;
        iy[i] = ys;
        cls[i] = isnan(g1) ? 0 : (int)g1;
    }
}

typedef struct { double lfc_p, lfc_t, el_p, el_t; int lfc_idx, el_idx, status; } lfc_el_out;

/* pf.py:1066-1198 */
static void lfc_el(int n, const double *p, const double *par, const double *env, double lcl_p, double lcl_t,
                   lfc_el_out *o) {
    double *ix = (double *)malloc(sizeof(double) * (size_t)n * 4);
    double *iy = ix + n, *jx = iy + n, *jy = jx + n;
    int *ic = (int *)malloc(sizeof(int) * (size_t)n * 2), *jc = ic + n;
    int m = n - 1;
    find_intersections(n, p, par, env, 1, ix, iy, ic);
    /* again ignoring the first level, re-labelled onto the same interval index (pf.py:1108-1112) */
    jx[0] = jy[0] = NAN; jc[0] = 0;
    if (n >= 2) find_intersections(n - 1, p + 1, par + 1, env + 1, 1, jx + 1, jy + 1, jc + 1);
    const double *sx = ix, *sy = iy; const int *sc = ic;
    if (!(env[0] != par[0])) { sx = jx; sy = jy; sc = jc; }              /* pf.py:1117-1120 */
    double lfc_p = NAN, lfc_t = NAN, el_p = NAN, el_t = NAN, any_inc = NAN;
    int lfc_idx = -1, el_idx = -1;
    for (int i = 0; i < m; i++) if (sc[i] > 0 && !isnan(sx[i])) {
        if (!(sx[i] <= any_inc)) any_inc = sx[i];
        if (sx[i] < lcl_p && !(sx[i] <= lfc_p)) { lfc_p = sx[i]; lfc_idx = i; }
    }
    for (int i = 0; i < m; i++) if (sc[i] > 0 && sx[i] == lfc_p && !isnan(sy[i]) && !(sy[i] <= lfc_t)) lfc_t = sy[i];
    for (int i = 0; i < m; i++) if (jc[i] < 0 && !isnan(jx[i]) && !(jx[i] >= el_p)) el_p = jx[i];
    for (int i = 0; i < m; i++) if (sc[i] < 0 && sx[i] == el_p && jc[i] < 0 && !isnan(jy[i]) && !(jy[i] <= el_t)) el_t = jy[i];
    for (int i = 0; i < m; i++) if (jc[i] < 0 && jx[i] == el_p) el_idx = i;
    /* top of the column where both temperatures exist (pf.py:1143-1155) */
    double ptop = NAN;
    for (int i = 0; i < n; i++) if (!isnan(par[i]) && !isnan(env[i]) && !isnan(p[i]) && !(p[i] >= ptop)) ptop = p[i];
    double top_par = NAN, top_env = NAN;
    for (int i = 0; i < n; i++) if (p[i] == ptop) {
        if (!isnan(par[i]) && !(par[i] <= top_par)) top_par = par[i];
        if (!isnan(env[i]) && !(env[i] <= top_env)) top_env = env[i];
    }
    o->status = 0;
    if (isnan(top_env) != isnan(nmax(env, n))) o->status |= 1;          /* assert pf.py:1149 */
    int el_exists = (top_par <= top_env) && (el_p < lcl_p);
    if (!el_exists) { el_p = NAN; el_t = NAN; el_idx = -1; }
    int lfc_missing = isnan(any_inc);
    int pos_parcel = 0;
    for (int i = 0; i < n; i++) if (p[i] < lcl_p && par[i] > env[i]) pos_parcel = 1;
    int replace = (pos_parcel && lfc_missing) || (!lfc_missing && isnan(lfc_p) && (el_p < lcl_p));
    if (replace) { lfc_p = lcl_p; lfc_t = lcl_t; lfc_idx = -2; }
    o->lfc_p = lfc_p; o->lfc_t = lfc_t; o->el_p = el_p; o->el_t = el_t; o->lfc_idx = lfc_idx; o->el_idx = el_idx;
    free(ix); free(ic);
}

/* pf.py:1291-1392 with helpers pf.py:164-206 (trapz) and pf.py:1200-1289 (trap_around_zeros) */
static void cape_cin_base(int n, const double *p, const double *env, double lfc_p, double el_p, const double *par,
                          int pos_neg, int post_zero, double *cape_out, double *cin_out) {
    if (isnan(el_p)) el_p = nmin(p, n);                                   /* pf.py:1329 */
    size_t N = (size_t)n;
    double *y = (double *)malloc(sizeof(double) * N * 9);
    double *X = y + N, *zx = X + N, *zy = zx + N, *zeros = zy + N;
    double *area_b = zeros + N, *x_b = area_b + N, *area_a = x_b + N, *x_a = area_a + N;
    int *zc = (int *)malloc(sizeof(int) * N);
    for (int i = 0; i < n; i++) { y[i] = par[i] - env[i]; X[i] = log(p[i]); zeros[i] = 0.0; }
    find_intersections(n, p, y, zeros, 1, zx, zy, zc);
    for (int i = 0; i < n; i++) { area_b[i] = x_b[i] = area_a[i] = x_a[i] = NAN; }
    for (int i = 0; i + 1 < n; i++) {
        if (isnan(zy[i])) continue;                                       /* pf.py:1241-1244 */
        double zlog = log(zx[i]);                                         /* pf.py:1237 */
        double dx = X[i] - zlog;                                          /* level just before the zero */
        area_b[i] = (y[i] / 2.0) * fabs(dx);
        x_b[i] = exp(X[i] - dx / 2.0);
        dx = X[i + 1] - zlog;                                             /* level just after the zero */
        area_a[i] = (y[i + 1] / 2.0) * fabs(dx);
        x_a[i] = exp(X[i + 1] - dx / 2.0);
    }
    double cape = 0, cin = 0;
    /* regular trapezoids on intervals without a valid zero (mask = isnan(area_before)) */
    for (int i = 0; i + 1 < n; i++) {
        if (!isnan(area_b[i])) continue;
        {   /* CAPE layer: both ends inside [el, lfc] (pf.py:1352-1353) */
            int in0 = (p[i] <= lfc_p) && (p[i] >= el_p), in1 = (p[i + 1] <= lfc_p) && (p[i + 1] >= el_p);
            if (in0 && in1) {
                double a = fabs(X[i + 1] - X[i]) * ((y[i] + y[i + 1]) / 2.0);
                if (!isnan(a) && (!pos_neg || a > 0)) cape += a;
            }
        }
        {   /* CIN layer: both ends at or below the LFC (pf.py:1371) */
            if ((p[i] >= lfc_p) && (p[i + 1] >= lfc_p)) {
                double a = fabs(X[i + 1] - X[i]) * ((y[i] + y[i + 1]) / 2.0);
                if (!isnan(a) && (!pos_neg || a < 0)) cin += a;
            }
        }
    }
    double cape_z = 0, cin_z = 0;
    for (int fam = 0; fam < 2; fam++) {
        const double *ar = fam ? area_a : area_b, *xx = fam ? x_a : x_b;
        for (int i = 0; i + 1 < n; i++) {
            if (isnan(ar[i])) continue;
            if (xx[i] <= lfc_p && xx[i] >= el_p && (!pos_neg || ar[i] > 0)) cape_z += ar[i];
            if (xx[i] >= lfc_p && (!pos_neg || ar[i] < 0)) cin_z += ar[i];
        }
    }
    cape = RD * cape + RD * cape_z;
    cin = RD * cin + RD * cin_z;
    if (post_zero && !(cin <= 0)) cin = 0;
    *cape_out = cape; *cin_out = cin;
    free(y); free(zc);
}

int xpo_lfc_el(int n, const double *p, const double *par, const double *env, double lcl_p, double lcl_t,
               double *out4, int *idx2) {
    lfc_el_out o; lfc_el(n, p, par, env, lcl_p, lcl_t, &o);
    out4[0] = o.lfc_p; out4[1] = o.lfc_t; out4[2] = o.el_p; out4[3] = o.el_t; idx2[0] = o.lfc_idx; idx2[1] = o.el_idx;
    return o.status;
}
void xpo_cape_cin_base(int n, const double *p, const double *env, double lfc_p, double el_p, const double *par,
                       int pos_neg, int post_zero, double *out2) {
    cape_cin_base(n, p, env, lfc_p, el_p, par, pos_neg, post_zero, &out2[0], &out2[1]);
}

/* ---- L2: parcel profile (pf.py:712-780) and LCL insertion (pf.py:806-931) ------------------- */
/* prof_* arrays of n; returns LCL triple */
static int parcel_profile(int n, const double *p, double pp, double pt, double ptd, int moist_mode,
                          double *t_par, double *tv_par, double *lcl3) {
    int st = xpo_lcl(pp, pt, ptd, &lcl3[0], &lcl3[1], &lcl3[2]);
    double w_parcel = mixing_ratio(pt, ptd, pp);
    double *above = (double *)malloc(sizeof(double) * (size_t)n);
    xpo_moist_lapse(n, p, lcl3[1], lcl3[0], moist_mode, above);
    for (int i = 0; i < n; i++) {
        double below = pt * pow(p[i] / pp, KAPPA);                          /* pf.py:313 */
        double t = (p[i] >= lcl3[0]) ? below : above[i];                    /* pf.py:767 */
        double w = (p[i] <= lcl3[0]) ? sat_mix(p[i], above[i]) : w_parcel;  /* pf.py:773 */
        t_par[i] = t;
        tv_par[i] = virt(t, w);
    }
    free(above);
    return st;
}
int xpo_parcel_profile(int n, const double *p, double pp, double pt, double ptd, int moist_mode,
                       double *t_par, double *tv_par, double *lcl3) {
    return parcel_profile(n, p, pp, pt, ptd, moist_mode, t_par, tv_par, lcl3);
}

/* Full profile with the LCL inserted.  out arrays of n+1: p, t_par, tv_par, t_env, tv_env, td_env */
static int profile_with_lcl(int n, const double *p, const double *t, const double *td, double pp, double pt, double ptd,
                            int log_interp, int moist_mode, double **out6, double *lcl3) {
    size_t N = (size_t)n;
    double *buf = (double *)malloc(sizeof(double) * N * 4);
    double *t_par = buf, *tv_par = buf + N, *tv_env = buf + 2 * N, *coords = buf + 3 * N;
    int st = parcel_profile(n, p, pp, pt, ptd, moist_mode, t_par, tv_par, lcl3);
    for (int i = 0; i < n; i++) tv_env[i] = virt(t[i], mixing_ratio(t[i], td[i], p[i]));     /* pf.py:839-843 */
    /* parcel side */
    {
        double *const d[3] = {(double *)p, t_par, tv_par};
        double level[3] = {lcl3[0], lcl3[1], lcl3[2]};
        double *o[3] = {out6[0], out6[1], out6[2]};
        insert_level(n, 3, d, level, o);
    }
    /* environment at the LCL (pf.py:897-920) */
    double at = log_interp ? log(lcl3[0]) : lcl3[0];
    for (int i = 0; i < n; i++) coords[i] = log_interp ? log(p[i]) : p[i];
    double t_l = linear_interp(n, coords, t, at), td_l = linear_interp(n, coords, td, at);
    double tv_l = virt(t_l, mixing_ratio(t_l, td_l, lcl3[0]));
    {
        double *pe = (double *)malloc(sizeof(double) * (N + 1));
        double *const d[4] = {(double *)p, (double *)t, tv_env, (double *)td};
        double level[4] = {lcl3[0], t_l, tv_l, td_l};
        double *o[4] = {pe, out6[3], out6[4], out6[5]};
        insert_level(n, 4, d, level, o);
        free(pe);
    }
    free(buf);
    return st;
}
int xpo_parcel_profile_with_lcl(int n, const double *p, const double *t, const double *td, double pp, double pt,
                                double ptd, int log_interp, int moist_mode, double *out /* 6*(n+1) */, double *lcl3) {
    double *o[6]; for (int v = 0; v < 6; v++) o[v] = out + (size_t)v * (size_t)(n + 1);
    return profile_with_lcl(n, p, t, td, pp, pt, ptd, log_interp, moist_mode, o, lcl3);
}

/* ---- front-ends: most-unstable (pf.py:63-135, 208-227, 1517-1555) and mixed layer (pf.py:137-289, 1604-1649) */
/* returns level index of the MU parcel or -1 */
int xpo_most_unstable_parcel(int n, const double *p, const double *t, const double *td, double depth, double *parcel3) {
    double bottom = nmax(p, n), bound = bottom - depth;
    double dmin = NAN, top = NAN;
    for (int i = 0; i < n; i++) { double d = fabs(p[i] - bound); if (!isnan(d) && !(d >= dmin)) dmin = d; }
    for (int i = 0; i < n; i++) if (fabs(p[i] - bound) == dmin && !(p[i] <= top)) top = p[i];    /* pf.py:224-226 */
    double best = NAN, pres = NAN;
    for (int i = 0; i < n; i++) {
        if (!(p[i] <= bottom && p[i] >= top)) continue;
        double e = theta_e(p[i], t[i], td[i]);
        if (!isnan(e) && !(e <= best)) best = e;
    }
    for (int i = 0; i < n; i++) {
        if (!(p[i] <= bottom && p[i] >= top)) continue;
        if (theta_e(p[i], t[i], td[i]) == best && !(p[i] <= pres)) pres = p[i];                   /* pf.py:128 */
    }
    int idx = -1;
    parcel3[0] = parcel3[1] = parcel3[2] = NAN;
    for (int i = 0; i < n; i++) if (p[i] == pres && p[i] <= bottom && p[i] >= top) {
        if (idx < 0) idx = i;
        if (!(p[i] <= parcel3[0])) parcel3[0] = p[i];
        if (!isnan(t[i]) && !(t[i] <= parcel3[1])) parcel3[1] = t[i];
        if (!isnan(td[i]) && !(td[i] <= parcel3[2])) parcel3[2] = td[i];
    }
    return idx;
}

/* layer mean by trapezoid in linear p over the lowest `depth` hPa with an interpolated top
   (get_layer pf.py:63-100 interpolate=True, mixed_layer pf.py:137-162) */
static double mixed_layer_mean(int n, const double *p, const double *v, double depth) {
    size_t N = (size_t)n;
    double *buf = (double *)calloc(3 * N + 2 * (N + 1), sizeof(double));
    double *lp = buf, *pi = buf + N, *vi = pi + N + 1, *dummy = vi + N + 1;
    (void)dummy;
    double bottom = nmax(p, n), top = bottom - depth;
    for (int i = 0; i < n; i++) lp[i] = log(p[i]);
    double lat = log(top);
    double p_l = linear_interp(n, lp, p, lat); (void)p_l;                    /* overwritten by top (pf.py:87) */
    double v_l = linear_interp(n, lp, v, lat);
    double *const d[2] = {(double *)p, (double *)v};
    double level[2] = {top, v_l};
    double *o[2] = {pi, vi};
    insert_level(n, 2, d, level, o);
    int m = n + 1;
    double pmin = NAN, pmax = NAN, s = 0;
    for (int i = 0; i < m; i++) {
        int in = (pi[i] <= bottom) && (pi[i] >= top);
        if (!in) { pi[i] = NAN; vi[i] = NAN; }
        else { if (!(pi[i] >= pmin)) pmin = pi[i]; if (!(pi[i] <= pmax)) pmax = pi[i]; }
    }
    for (int i = 0; i + 1 < m; i++) {
        double a = fabs(pi[i + 1] - pi[i]) * ((vi[i] + vi[i + 1]) / 2.0);
        if (!isnan(a)) s += a;
    }
    double r = (1.0 / fabs(pmin - pmax)) * s;
    free(buf);
    return r;
}
double xpo_mixed_layer(int n, const double *p, const double *v, double depth) { return mixed_layer_mean(n, p, v, depth); }

void xpo_mixed_parcel(int n, const double *p, const double *t, const double *td, double depth, double *parcel3) {
    double *buf = (double *)malloc(sizeof(double) * (size_t)n * 2);
    double *theta = buf, *w = buf + n;
    for (int i = 0; i < n; i++) { theta[i] = t[i] / pow(p[i] / 1000.0, KAPPA); w[i] = sat_mix(p[i], td[i]); }
    double th = mixed_layer_mean(n, p, theta, depth), wm = mixed_layer_mean(n, p, w, depth);
    double p0 = p[0];                                                       /* pf.py:250 */
    parcel3[0] = p0;
    parcel3[1] = th * pow(p0 / 1000.0, KAPPA);
    parcel3[2] = dewpoint_of_e(vapor_pressure(p0, wm));
    free(buf);
}

/* ---- L3 driver: cape_cin (pf.py:1394-1475) for one column --------------------------------------- */
typedef struct {
    double cape, cin, lcl_p, lcl_t, lcl_tv, lfc_p, lfc_t, el_p, el_t;
    int32_t lfc_idx, el_idx, status, parcel_idx;
} xpo_col_out;

static void cape_cin_column(int n, const double *p, const double *t, const double *td, double pp, double pt, double ptd,
                            const xpo_opts *o, xpo_col_out *r, double *profile6 /* nullable, 6*(n+1) */) {
    size_t M = (size_t)n + 1;
    double *buf = profile6 ? profile6 : (double *)malloc(sizeof(double) * M * 6);
    double *out6[6]; for (int v = 0; v < 6; v++) out6[v] = buf + (size_t)v * M;
    double lcl3[3];
    int st = profile_with_lcl(n, p, t, td, pp, pt, ptd, o->lcl_interp_log, o->moist_mode, out6, lcl3);
    const double *par = o->vtc ? out6[2] : out6[1], *env = o->vtc ? out6[4] : out6[3];
    double lt = o->vtc ? lcl3[2] : lcl3[1];
    lfc_el_out le; lfc_el(n + 1, out6[0], par, env, lcl3[0], lt, &le);
    cape_cin_base(n + 1, out6[0], env, le.lfc_p, le.el_p, par, o->pos_cape_neg_cin, o->post_zero_cin, &r->cape, &r->cin);
    r->lcl_p = lcl3[0]; r->lcl_t = lcl3[1]; r->lcl_tv = lcl3[2];
    r->lfc_p = le.lfc_p; r->lfc_t = le.lfc_t; r->el_p = le.el_p; r->el_t = le.el_t;
    r->lfc_idx = le.lfc_idx; r->el_idx = le.el_idx;
    r->status = (st ? 2 : 0) | le.status;
    if (!profile6) free(buf);
}

/* Grid entry point.  Arrays are (nlev, ncol) with element (k, c) at k*lev_stride + c*col_stride.
   parcel: for parcel_mode 3 a (3, ncol) array p,T,Td (contiguous rows); else NULL.
   profile: nullable, (6, nlev+1, ncol) contiguous; for MU/ML the profile holds the re-based column
   (levels removed from the bottom, NaN padding at the top).
   scalars: (9, ncol) doubles: cape cin lcl_p lcl_t lcl_tv lfc_p lfc_t el_p el_t ; ints: (4, ncol) */
int xpo_cape_cin(const double *p, const double *t, const double *td, int64_t nlev, int64_t ncol, int64_t lev_stride,
                 int64_t col_stride, const double *parcel, const xpo_opts *o, double *scalars, int32_t *ints,
                 double *profile, int nthreads) {
    int n = (int)nlev;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        size_t N = (size_t)n + 2;
        double *cp = (double *)malloc(sizeof(double) * N * 3), *ct = cp + N, *ctd = ct + N;
        double *prof = profile ? (double *)malloc(sizeof(double) * 6 * N) : NULL;
#pragma omp for schedule(dynamic, 64)
        for (int64_t c = 0; c < ncol; c++) {
            for (int k = 0; k < n; k++) {
                int64_t off = (int64_t)k * lev_stride + c * col_stride;
                cp[k] = p[off]; ct[k] = t[off]; ctd[k] = td[off];
            }
            double par3[3]; int pidx = 0; int m = n;
            const double *xp = cp, *xt = ct, *xtd = ctd;
            if (o->parcel_mode == 0) { par3[0] = cp[0]; par3[1] = ct[0]; par3[2] = ctd[0]; }
            else if (o->parcel_mode == 3) { par3[0] = parcel[c]; par3[1] = parcel[ncol + c]; par3[2] = parcel[2 * ncol + c]; pidx = -1; }
            else if (o->parcel_mode == 1) {
                pidx = xpo_most_unstable_parcel(n, cp, ct, ctd, o->depth, par3);
                /* keep p <= p_MU, shift down (pf.py:1551-1553); NaN padding on top */
                int w = 0;
                for (int k = 0; k < n; k++) if (cp[k] <= par3[0]) { cp[w] = cp[k]; ct[w] = ct[k]; ctd[w] = ctd[k]; w++; }
                for (int k = w; k < n; k++) cp[k] = ct[k] = ctd[k] = NAN;
            } else {
                xpo_mixed_parcel(n, cp, ct, ctd, o->depth, par3);
                double lim = nmax(cp, n) - o->depth;
                int w = 1;
                /* [parcel] + levels with p < p_max - depth (pf.py:1636-1644) */
                double *tp = (double *)malloc(sizeof(double) * N * 3), *tt = tp + N, *ttd = tt + N;
                tp[0] = par3[0]; tt[0] = par3[1]; ttd[0] = par3[2];
                for (int k = 0; k < n; k++) if (cp[k] < lim) { tp[w] = cp[k]; tt[w] = ct[k]; ttd[w] = ctd[k]; w++; }
                for (int k = w; k < n + 1; k++) tp[k] = tt[k] = ttd[k] = NAN;
                m = n + 1;
                memcpy(cp, tp, sizeof(double) * (size_t)m); memcpy(ct, tt, sizeof(double) * (size_t)m);
                memcpy(ctd, ttd, sizeof(double) * (size_t)m);
                free(tp);
                pidx = -1;
            }
            /* trailing NaN-pressure padding is inert in the reference; trim it so the
               array helpers see the populated part only */
            int mm = m; while (mm > 1 && isnan(xp[mm - 1]) && isnan(xt[mm - 1]) && isnan(xtd[mm - 1])) mm--;
            xpo_col_out r;
            cape_cin_column(mm, xp, xt, xtd, par3[0], par3[1], par3[2], o, &r, prof);
            r.parcel_idx = pidx;
            double *s = scalars + c;
            s[0] = r.cape; s[ncol] = r.cin; s[2 * ncol] = r.lcl_p; s[3 * ncol] = r.lcl_t; s[4 * ncol] = r.lcl_tv;
            s[5 * ncol] = r.lfc_p; s[6 * ncol] = r.lfc_t; s[7 * ncol] = r.el_p; s[8 * ncol] = r.el_t;
            if (ints) { ints[c] = r.lfc_idx; ints[ncol + c] = r.el_idx; ints[2 * ncol + c] = r.status; ints[3 * ncol + c] = r.parcel_idx; }
            if (profile) {
                int64_t M = nlev + 1;
                for (int v = 0; v < 6; v++)
                    for (int64_t k = 0; k < M; k++)
                        profile[((int64_t)v * M + k) * ncol + c] = (k < mm + 1) ? prof[(size_t)v * (size_t)(mm + 1) + (size_t)k] : NAN;
            }
        }
        free(cp); if (prof) free(prof);
    }
    return 0;
}

int xpo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
