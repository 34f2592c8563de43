/*
 * oracle/lsqr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's LSQR hot path
 * (jacobwilliams/LSQR).  It exists only so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg have something to check the HIP path against
 * (and to time beside it).  Nothing under lsqr_amd/ may include, link, import
 * or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here bit-for-bit against fixtures in tests/golden/ that were produced by the
 * unmodified reference compiled from /root/reference/src (oracle/Makefile ->
 * oracle/_ref/libref_lsqr.so, generator tests/golden/gen_golden.py), and, when
 * oracle/_ref is present, against the live reference library itself.
 * Build both sides with -ffp-contract=off so the arithmetic is comparable.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference).
 */
#include "lsqr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------- */
/* BLAS-1 (src/lsqrblas.f90)                                              */
/* ---------------------------------------------------------------------- */

/* src/lsqrblas.f90:25-67  dcopy (the unroll-by-7 is order-irrelevant). */
void oracle_dcopy(int n, const double *dx, int incx, double *dy, int incy)
{
    if (n <= 0) return;
    if (incx == 1 && incy == 1) {
        memcpy(dy, dx, (size_t)n * sizeof(double));
        return;
    }
    long ix = 0, iy = 0;
    if (incx < 0) ix = (long)(-n + 1) * incx;
    if (incy < 0) iy = (long)(-n + 1) * incy;
    for (int i = 0; i < n; ++i) {
        dy[iy] = dx[ix];
        ix += incx;
        iy += incy;
    }
}

/* src/lsqrblas.f90:74-116  ddot: clean-up loop of mod(n,5) elements first,
 * then groups of five added left to right into one running sum. */
double oracle_ddot(int n, const double *dx, int incx, const double *dy, int incy)
{
    double dtemp = 0.0;
    if (n <= 0) return 0.0;
    if (incx == 1 && incy == 1) {
        int m = n % 5;
        for (int i = 0; i < m; ++i) dtemp = dtemp + dx[i] * dy[i];
        if (n < 5) return dtemp;
        for (int i = m; i < n; i += 5) {
            dtemp = dtemp + dx[i] * dy[i] + dx[i + 1] * dy[i + 1] + dx[i + 2] * dy[i + 2] +
                    dx[i + 3] * dy[i + 3] + dx[i + 4] * dy[i + 4];
        }
        return dtemp;
    }
    long ix = 0, iy = 0;
    if (incx < 0) ix = (long)(-n + 1) * incx;
    if (incy < 0) iy = (long)(-n + 1) * incy;
    for (int i = 0; i < n; ++i) {
        dtemp = dtemp + dx[ix] * dy[iy];
        ix += incx;
        iy += incy;
    }
    return dtemp;
}

/* 0 = the reference's dnrm2 (everything pinned against the reference uses it).  1 = sqrt of a pairwise
 * sum of squares: a legal, more accurate evaluation of the same norm, used ONLY by
 * tests/golden/gen_lstp_band.py to measure how the iteration counts of the 18-problem suite respond to
 * the accuracy of the sums (the GPU sums in trees). */
static int g_norm_order = 0;
void oracle_set_norm_order(int order) { g_norm_order = order; }

/* Test hook (never set by the parity tests themselves): move ONE norm of the bidiagonalisation -- beta (which = 1)
 * or alpha (which = 2) of iteration `itn` (0: the start, src/lsqr.f90:632-641) -- by `ulps` units in the last place.
 * tests/fuzz_layouts.py measures with it how far x of the reference itself moves under the smallest possible change
 * of a norm -- the one rounding a permutation of the COO input does not touch -- and holds the GPU to a multiple of
 * that.  which = 0 switches the hook off. */
static int g_ulp_itn = 0, g_ulp_which = 0, g_ulp_count = 0;
void oracle_set_norm_ulp(int itn, int which, int ulps)
{
    g_ulp_itn = itn;
    g_ulp_which = which;
    g_ulp_count = ulps;
}
static double nudge(double v, int itn, int which)
{
    if (g_ulp_which != which || g_ulp_itn != itn || v == 0.0) return v;
    for (int k = 0; k < (g_ulp_count < 0 ? -g_ulp_count : g_ulp_count); ++k)
        v = nextafter(v, g_ulp_count > 0 ? INFINITY : -INFINITY);
    return v;
}

static double pairwise_sumsq(const double *x, int n)
{
    if (n <= 8) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = x[i] * x[i] + s;
        return s;
    }
    const int h = n / 2;
    return pairwise_sumsq(x, h) + pairwise_sumsq(x + h, n - h);
}

/* src/lsqrblas.f90:123-159  dnrm2: scaled sum of squares (dlassq recurrence). */
double oracle_dnrm2(int n, const double *x, int incx)
{
    if (n < 1 || incx < 1) return 0.0;
    if (n == 1) return fabs(x[0]);
    if (g_norm_order == 1 && incx == 1) return sqrt(pairwise_sumsq(x, n));
    double scale = 0.0, ssq = 1.0;
    for (long ix = 0; ix <= (long)(n - 1) * incx; ix += incx) {
        if (x[ix] != 0.0) {
            double absxi = fabs(x[ix]);
            if (scale < absxi) {
                double r = scale / absxi;
                ssq = 1.0 + ssq * (r * r);
                scale = absxi;
            } else {
                double r = absxi / scale;
                ssq = ssq + r * r;
            }
        }
    }
    return scale * sqrt(ssq);
}

/* src/lsqrblas.f90:166-201  dscal. */
void oracle_dscal(int n, double da, double *dx, int incx)
{
    if (n <= 0 || incx <= 0) return;
    if (incx == 1) {
        for (int i = 0; i < n; ++i) dx[i] = da * dx[i];
        return;
    }
    long nincx = (long)n * incx;
    for (long i = 0; i < nincx; i += incx) dx[i] = da * dx[i];
}

/* src/lsqr.f90:1164-1179  d2norm. */
double oracle_d2norm(double a, double b)
{
    double scale = fabs(a) + fabs(b);
    if (scale == 0.0) return 0.0;
    double p = a / scale, q = b / scale;
    return scale * sqrt(p * p + q * q);
}

/* ---------------------------------------------------------------------- */
/* aprod_ez (src/lsqr.f90:134-200)                                        */
/* ---------------------------------------------------------------------- */

/* mode 1: y += A x  (row sums formed from zero in COO order, then added, :166-174)
 * mode 2: x += A' y (:186-194).  irow/icol are 1-based like the reference.
 * scratch must hold max(m,n) doubles.  Returns 0, or 5 for a bad mode (:197). */
/* Test hook: 1 = every row sum of aprod as a COMPENSATED sum (Neumaier: the error of each add is carried along, so the
 * result is the correctly rounded sum for all practical purposes) instead of the reference's left-to-right one.
 * Never set by a parity test: tests/fuzz_layouts.py measures with it (and oracle_set_norm_order(1)) how far the
 * reference's x lies from an evaluation of the same recurrences with accurate sums -- its own rounding error,
 * including the part no permutation of the input reveals (thousands of EQUAL addends in a row drift one way in
 * every order: 64 x 5000 with one 6000-entry row of dictionary values, 2.8e-14 of the row sum in all of them). */
static int g_accurate_rowsums = 0;
void oracle_set_accurate_rowsums(int on) { g_accurate_rowsums = on; }

static void comp_add(double *s, double *c, double t)
{
    const double sum = *s + t;
    if (fabs(*s) >= fabs(t)) *c += (*s - sum) + t;
    else *c += (t - sum) + *s;
    *s = sum;
}

int oracle_aprod(int mode, int m, int n, long long nnz, const int *irow, const int *icol,
                 const double *a, double *x, double *y, double *scratch)
{
    if (g_accurate_rowsums && (mode == 1 || mode == 2)) {
        const int len = mode == 1 ? m : n;
        double *comp = (double *)calloc((size_t)(len > 0 ? len : 1), sizeof(double));
        if (!comp) return 5;
        for (int i = 0; i < len; ++i) scratch[i] = 0.0;
        for (long long k = 0; k < nnz; ++k) {
            const int r = irow[k] - 1, c = icol[k] - 1;
            if (mode == 1) comp_add(&scratch[r], &comp[r], a[k] * x[c]);
            else comp_add(&scratch[c], &comp[c], a[k] * y[r]);
        }
        if (mode == 1) for (int i = 0; i < m; ++i) y[i] = y[i] + (scratch[i] + comp[i]);
        else for (int j = 0; j < n; ++j) x[j] = x[j] + (scratch[j] + comp[j]);
        free(comp);
        return 0;
    }
    if (mode == 1) {
        for (int i = 0; i < m; ++i) scratch[i] = 0.0;
        for (long long k = 0; k < nnz; ++k) {
            int r = irow[k] - 1, c = icol[k] - 1;
            scratch[r] = scratch[r] + a[k] * x[c];
        }
        for (int i = 0; i < m; ++i) y[i] = y[i] + scratch[i];
        return 0;
    }
    if (mode == 2) {
        for (int j = 0; j < n; ++j) scratch[j] = 0.0;
        for (long long k = 0; k < nnz; ++k) {
            int r = irow[k] - 1, c = icol[k] - 1;
            scratch[c] = scratch[c] + a[k] * y[r];
        }
        for (int j = 0; j < n; ++j) x[j] = x[j] + scratch[j];
        return 0;
    }
    return 5;
}

/* initialize_ez validation (src/lsqr.f90:109-111).  Only upper bounds are
 * checked by the reference; lower bounds are not.  Returns the reference's
 * error ordinal: 0 ok, 2 'invalid irow or m', 3 'invalid icol or n'. */
int oracle_validate(int m, int n, long long nnz, const int *irow, const int *icol)
{
    for (long long k = 0; k < nnz; ++k)
        if (irow[k] > m) return 2;
    for (long long k = 0; k < nnz; ++k)
        if (icol[k] > n) return 3;
    return 0;
}

/* ---------------------------------------------------------------------- */
/* The operator interface of the abstract class (src/lsqr.f90:16-30, 67-82):  */
/* aprod(mode, m, n, x, y) -- mode 1: y += A x, mode 2: x += A' y.            */
/* ---------------------------------------------------------------------- */
struct coo_ctx {
    long long nnz;
    const int *irow, *icol;
    const double *a;
    double *scratch; /* max(m, n) */
};
static void coo_aprod(void *vctx, int mode, int m, int n, double *x, double *y)
{
    struct coo_ctx *c = (struct coo_ctx *)vctx;
    oracle_aprod(mode, m, n, c->nnz, c->irow, c->icol, c->a, x, y, c->scratch);
}

/* ---------------------------------------------------------------------- */
/* LSQR (src/lsqr.f90:432-882) through solve_ez (src/lsqr.f90:207-259)     */
/* ---------------------------------------------------------------------- */

int oracle_lsqr_op(int m, int n, oracle_aprod_fn aprod, void *ctx,
                   const double *b, double damp, double atol, double btol, double conlim,
                   int itnlim, int wantse, double *x, double *se, int *istop_out, int *itn_out,
                   double *anorm_out, double *acond_out, double *rnorm_out, double *arnorm_out,
                   double *xnorm_out, double *log, int logcap)
{
    double *u = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    double *v = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double *w = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!u || !v || !w) {
        free(u); free(v); free(w);
        return 1;
    }
    memcpy(u, b, sizeof(double) * (size_t)m); /* solve_ez :242 */

    /* :597-617 */
    const int damped = damp > 0.0;
    int itn = 0, istop = 0, nstop = 0;
    double ctol = conlim > 0.0 ? 1.0 / conlim : 0.0;
    double anorm = 0.0, acond = 0.0, dnorm = 0.0, dxmax = 0.0, res2 = 0.0, psi = 0.0;
    double xnorm = 0.0, xnorm1 = 0.0, cs2 = -1.0, sn2 = 0.0, z = 0.0;
    double rnorm = 0.0, arnorm = 0.0, bnorm = 0.0;
    double rhobar = 0.0, phibar = 0.0;
    int maxdx = 0;
    (void)maxdx;

    /* :621-630 */
    for (int i = 0; i < n; ++i) { v[i] = 0.0; x[i] = 0.0; }
    if (wantse) for (int i = 0; i < n; ++i) se[i] = 0.0;

    /* :632-644 */
    double alpha = 0.0;
    double beta = nudge(oracle_dnrm2(m, u, 1), 0, 1);
    if (beta > 0.0) {
        oracle_dscal(m, 1.0 / beta, u, 1);
        aprod(ctx, 2, m, n, v, u);
        alpha = nudge(oracle_dnrm2(n, v, 1), 0, 2);
    }
    if (alpha > 0.0) {
        oracle_dscal(n, 1.0 / alpha, v, 1);
        oracle_dcopy(n, v, 1, w, 1);
    }

    /* :646-653.  The reference leaves rnorm/bnorm unassigned when the loop is
     * skipped; here they are defined as beta. */
    arnorm = alpha * beta;
    bnorm = beta;
    rnorm = beta;

    if (arnorm != 0.0) {
        rhobar = alpha;
        phibar = beta;

        for (;;) {
            itn = itn + 1; /* :675 */

            /* :681-683 */
            oracle_dscal(m, -alpha, u, 1);
            aprod(ctx, 1, m, n, v, u);
            beta = nudge(oracle_dnrm2(m, u, 1), itn, 1);

            /* :687-689 */
            double temp = oracle_d2norm(alpha, beta);
            temp = oracle_d2norm(temp, damp);
            anorm = oracle_d2norm(anorm, temp);

            /* :691-699 */
            if (beta > 0.0) {
                oracle_dscal(m, 1.0 / beta, u, 1);
                oracle_dscal(n, -beta, v, 1);
                aprod(ctx, 2, m, n, v, u);
                alpha = nudge(oracle_dnrm2(n, v, 1), itn, 2);
                if (alpha > 0.0) oracle_dscal(n, 1.0 / alpha, v, 1);
            }

            /* :703-710 */
            double rhbar1 = rhobar;
            if (damped) {
                rhbar1 = oracle_d2norm(rhobar, damp);
                double cs1 = rhobar / rhbar1;
                double sn1 = damp / rhbar1;
                psi = sn1 * phibar;
                phibar = cs1 * phibar;
            }

            /* :714-721 */
            double rho = oracle_d2norm(rhbar1, beta);
            double cs = rhbar1 / rho;
            double sn = beta / rho;
            double theta = sn * alpha;
            rhobar = -cs * alpha;
            double phi = cs * phibar;
            phibar = sn * phibar;
            double tau = sn * phi;

            /* :724-745 */
            double t1 = phi / rho;
            double t2 = -theta / rho;
            double t3 = 1.0 / rho;
            double dknorm = 0.0;
            if (wantse) {
                for (int i = 0; i < n; ++i) {
                    double t = w[i];
                    x[i] = t1 * t + x[i];
                    w[i] = t2 * t + v[i];
                    t = (t3 * t) * (t3 * t);
                    se[i] = t + se[i];
                    dknorm = t + dknorm;
                }
            } else {
                for (int i = 0; i < n; ++i) {
                    double t = w[i];
                    x[i] = t1 * t + x[i];
                    w[i] = t2 * t + v[i];
                    dknorm = (t3 * t) * (t3 * t) + dknorm;
                }
            }

            /* :751-757 */
            dknorm = sqrt(dknorm);
            dnorm = oracle_d2norm(dnorm, dknorm);
            double dxk = fabs(phi * dknorm);
            if (dxmax < dxk) { dxmax = dxk; maxdx = itn; }

            /* :762-771 */
            double delta = sn2 * rho;
            double gambar = -cs2 * rho;
            double rhs = phi - delta * z;
            double zbar = rhs / gambar;
            xnorm = oracle_d2norm(xnorm1, zbar);
            double gamma = oracle_d2norm(gambar, theta);
            cs2 = gambar / gamma;
            sn2 = theta / gamma;
            z = rhs / gamma;
            xnorm1 = oracle_d2norm(xnorm1, z);

            /* :776-790 */
            acond = anorm * dnorm;
            res2 = oracle_d2norm(res2, psi);
            rnorm = oracle_d2norm(res2, phibar);
            arnorm = alpha * fabs(tau);

            double alfopt = sqrt(rnorm / (dnorm * xnorm));
            double test1 = rnorm / bnorm;
            double test2 = 0.0;
            if (rnorm > 0.0) test2 = arnorm / (anorm * rnorm);
            double test3 = 1.0 / acond;
            t1 = test1 / (1.0 + anorm * xnorm / bnorm);
            double rtol = btol + atol * anorm * xnorm / bnorm;

            /* :798-810  (later assignments win: order 5,4,2,1 then 4,2,1) */
            t3 = 1.0 + test3;
            t2 = 1.0 + test2;
            t1 = 1.0 + t1;
            if (itn >= itnlim) istop = 5;
            if (t3 <= 1.0) istop = 4;
            if (t2 <= 1.0) istop = 2;
            if (t1 <= 1.0) istop = 1;
            if (test3 <= ctol) istop = 4;
            if (test2 <= atol) istop = 2;
            if (test1 <= rtol) istop = 1;

            /* per-iteration record: the values the reference prints at :828-829 */
            if (log && itn <= logcap) {
                double *r = log + (size_t)(itn - 1) * ORACLE_LOG_STRIDE;
                r[0] = (double)itn; r[1] = n > 0 ? x[0] : 0.0; r[2] = rnorm; r[3] = test1;
                r[4] = test2; r[5] = anorm; r[6] = acond; r[7] = phi; r[8] = dknorm;
                r[9] = dxk; r[10] = alfopt; r[11] = (double)istop; r[12] = rtol; r[13] = xnorm;
            }

            /* :843-850 */
            if (istop == 0) {
                nstop = 0;
            } else {
                const int nconv = 1;
                nstop = nstop + 1;
                if (nstop < nconv && itn < itnlim) istop = 0;
            }
            if (istop != 0) break;
        }

        /* :857-865 */
        if (wantse) {
            double t = 1.0;
            if (m > n) t = (double)(m - n);
            if (damped) t = (double)m;
            t = rnorm / sqrt(t);
            for (int i = 0; i < n; ++i) se[i] = t * sqrt(se[i]);
        }
    }

    if (damped && istop == 2) istop = 3; /* :871 */

    *istop_out = istop;
    if (itn_out) *itn_out = itn;
    if (anorm_out) *anorm_out = anorm;
    if (acond_out) *acond_out = acond;
    if (rnorm_out) *rnorm_out = rnorm;
    if (arnorm_out) *arnorm_out = arnorm;
    if (xnorm_out) *xnorm_out = xnorm;
    free(u); free(v); free(w);
    return 0;
}

/* solve_ez (src/lsqr.f90:207-259): LSQR on the COO operator */
int oracle_lsqr_ez(int m, int n, long long nnz, const int *irow, const int *icol, const double *a,
                   const double *b, double damp, double atol, double btol, double conlim,
                   int itnlim, int wantse, double *x, double *se, int *istop_out, int *itn_out,
                   double *anorm_out, double *acond_out, double *rnorm_out, double *arnorm_out,
                   double *xnorm_out, double *log, int logcap)
{
    int mx = m > n ? m : n;
    struct coo_ctx c = {nnz, irow, icol, a, (double *)malloc(sizeof(double) * (size_t)(mx > 0 ? mx : 1))};
    if (!c.scratch) return 1;
    int rc = oracle_lsqr_op(m, n, coo_aprod, &c, b, damp, atol, btol, conlim, itnlim, wantse, x, se, istop_out,
                            itn_out, anorm_out, acond_out, rnorm_out, arnorm_out, xnorm_out, log, logcap);
    free(c.scratch);
    return rc;
}

/* ---------------------------------------------------------------------- */
/* acheck (src/lsqr.f90:908-994)                                          */
/* ---------------------------------------------------------------------- */
int oracle_acheck_op(int m, int n, oracle_aprod_fn aprod, void *ctx, double eps, double *err_out)
{
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    double *w = (double *)malloc(sizeof(double) * (size_t)m);
    double *x = (double *)malloc(sizeof(double) * (size_t)n);
    double *y = (double *)malloc(sizeof(double) * (size_t)m);
    const double tol = pow(eps, 0.5); /* :939 */
    double t = 1.0;
    for (int j = 0; j < n; ++j) { t = t + 1.0; x[j] = sqrt(t); }          /* :946-950 */
    t = 1.0;
    for (int i = 0; i < m; ++i) { t = t + 1.0; y[i] = 1.0 / sqrt(t); }    /* :952-956 */
    double alfa = oracle_dnrm2(n, x, 1);
    double beta = oracle_dnrm2(m, y, 1);
    oracle_dscal(n, 1.0 / alfa, x, 1);
    oracle_dscal(m, 1.0 / beta, y, 1);
    oracle_dcopy(m, y, 1, w, 1);                                          /* :969-972 */
    oracle_dcopy(n, x, 1, v, 1);
    aprod(ctx, 1, m, n, x, w);
    aprod(ctx, 2, m, n, v, y);
    alfa = oracle_ddot(m, y, 1, w, 1);                                    /* :976-980 */
    beta = oracle_ddot(n, x, 1, v, 1);
    double test1 = fabs(alfa - beta);
    double test2 = 1.0 + fabs(alfa) + fabs(beta);
    double test3 = test1 / test2;
    if (err_out) *err_out = test3;
    free(v); free(w); free(x); free(y);
    return test3 <= tol ? 0 : 1;                                          /* :984-992 */
}

int oracle_acheck(int m, int n, long long nnz, const int *irow, const int *icol, const double *a,
                  double eps, double *err_out)
{
    int mx = m > n ? m : n;
    struct coo_ctx c = {nnz, irow, icol, a, (double *)malloc(sizeof(double) * (size_t)(mx > 0 ? mx : 1))};
    int inform = oracle_acheck_op(m, n, coo_aprod, &c, eps, err_out);
    free(c.scratch);
    return inform;
}

/* ---------------------------------------------------------------------- */
/* xcheck (src/lsqr.f90:1015-1154)                                        */
/* ---------------------------------------------------------------------- */
int oracle_xcheck_op(int m, int n, oracle_aprod_fn aprod, void *ctx,
                     double anorm, double damp, double eps, const double *b, const double *x,
                     double *u, double *v, double *w, double *tests /* [3] */)
{
    double *xtmp = (double *)malloc(sizeof(double) * (size_t)n);
    const double dampsq = damp * damp;
    const double tol = pow(eps, 0.5);
    memcpy(xtmp, x, sizeof(double) * (size_t)n);

    oracle_dcopy(m, b, 1, u, 1);                                          /* :1073-1076 */
    oracle_dscal(m, -1.0, u, 1);
    aprod(ctx, 1, m, n, xtmp, u);
    oracle_dscal(m, -1.0, u, 1);
    for (int j = 0; j < n; ++j) v[j] = 0.0;                               /* :1080-1083 */
    aprod(ctx, 2, m, n, v, u);
    oracle_dcopy(n, v, 1, w, 1);                                          /* :1089-1094 */
    if (damp != 0.0)
        for (int j = 0; j < n; ++j) w[j] = w[j] - dampsq * x[j];

    double bnorm = oracle_dnrm2(m, b, 1);                                 /* :1098-1101 */
    double xnorm = oracle_dnrm2(n, x, 1);
    double rho1 = oracle_dnrm2(m, u, 1);
    double sigma1 = oracle_dnrm2(n, v, 1);
    double rho2, sigma2;
    if (damp == 0.0) {                                                    /* :1110-1124 */
        rho2 = rho1;
        sigma2 = sigma1;
    } else {
        rho2 = sqrt(rho1 * rho1 + dampsq * (xnorm * xnorm));
        sigma2 = oracle_dnrm2(n, w, 1);
    }
    int inform;
    double test1, test2, test3;
    if (bnorm == 0.0 && xnorm == 0.0) {                                   /* :1129-1144 */
        inform = 0; test1 = test2 = test3 = 0.0;
    } else {
        inform = 4;
        test1 = rho1 / (bnorm + anorm * xnorm);
        test2 = 0.0;
        if (rho1 > 0.0) test2 = sigma1 / (anorm * rho1);
        test3 = test2;
        if (rho2 > 0.0) test3 = sigma2 / (anorm * rho2);
        if (test3 <= tol) inform = 3;
        if (test2 <= tol) inform = 2;
        if (test1 <= tol) inform = 1;
    }
    tests[0] = test1; tests[1] = test2; tests[2] = test3;
    free(xtmp);
    return inform;
}

int oracle_xcheck(int m, int n, long long nnz, const int *irow, const int *icol, const double *a,
                  double anorm, double damp, double eps, const double *b, const double *x,
                  double *u, double *v, double *w, double *tests /* [3] */)
{
    int mx = m > n ? m : n;
    struct coo_ctx c = {nnz, irow, icol, a, (double *)malloc(sizeof(double) * (size_t)(mx > 0 ? mx : 1))};
    int inform = oracle_xcheck_op(m, n, coo_aprod, &c, anorm, damp, eps, b, x, u, v, w, tests);
    free(c.scratch);
    return inform;
}
