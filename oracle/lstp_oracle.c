/* oracle/lstp_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the reference's own test-problem class (Paige & Saunders' generator):
 * the operator A = HY * D * HZ (two Householder reflections around a diagonal), the problem
 * generator `lstp`, and the driver `test` that the reference's 18-problem suite runs.
 *
 *   hprod        test/lsqrtest_module.f90:385-403
 *   aprod1       test/lsqrtest_module.f90:319-343     y += A x
 *   aprod2       test/lsqrtest_module.f90:353-377     x += A' y
 *   lstp         test/lsqrtest_module.f90:422-505
 *   test         test/lsqrtest_module.f90:119-272     (acheck -> lsqr -> xcheck -> error in x)
 *   lsqr_test    test/lsqrtest_module.f90:55-94       the 18 (m, n, npower, damp) combinations
 *
 * Sequential sums in the reference's order (s = hz(i)*x(i) + s), so on the same compiler
 * flags this is the reference bit for bit.  Pinned against the log the compiled reference
 * writes here (oracle/_ref/lsqrtest -> tests/golden/LSQR_ref.LIS, tests/test_lstp_oracle.py)
 * and against the log the reference ships (test/LSQR.LIS facts quoted in that test).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "lsqr_oracle.h"

/* Order of the dot product inside hprod.  0 = the reference's (ascending, sequential): everything
 * pinned against the reference uses it.  1 = descending, 2 = pairwise tree, 3 = eight interleaved
 * partial sums: legal re-orderings of the same sum, used ONLY to measure how far the iteration counts of
 * the 18-problem suite move under rounding (tests/golden/gen_lstp_band.py) -- the band a GPU run with
 * tree sums is then held to. */
static int g_sum_order = 0;
void oracle_lstp_set_sum_order(int order) { g_sum_order = order; }

static double pairwise_dot(const double *a, const double *b, int n)
{
    if (n <= 8) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = a[i] * b[i] + s;
        return s;
    }
    const int h = n / 2;
    return pairwise_dot(a, b, h) + pairwise_dot(a + h, b + h, n - h);
}

/* y = (I - 2 hz hz') x                              test/lsqrtest_module.f90:385-403 */
void oracle_hprod(int n, const double *hz, const double *x, double *y)
{
    double s = 0.0;
    if (g_sum_order == 1) {
        for (int i = n - 1; i >= 0; --i) s = hz[i] * x[i] + s;
    } else if (g_sum_order == 2) {
        s = pairwise_dot(hz, x, n);
    } else if (g_sum_order == 3) {
        double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) p[i & 7] = hz[i] * x[i] + p[i & 7];
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    } else {
        for (int i = 0; i < n; ++i) s = hz[i] * x[i] + s;
    }
    s = s + s;
    for (int i = 0; i < n; ++i) y[i] = x[i] - s * hz[i];
}

/* mode 1: y += A x (:319-343); mode 2: x += A' y (:353-377) */
void oracle_lstp_aprod(void *vctx, int mode, int m, int n, double *x, double *y)
{
    oracle_lstp_t *c = (oracle_lstp_t *)vctx;
    const int minmn = m < n ? m : n;
    double *w = c->w;
    if (mode == 1) {
        oracle_hprod(n, c->hz, x, w);
        for (int i = 0; i < minmn; ++i) w[i] = c->d[i] * w[i];
        for (int i = n; i < m; ++i) w[i] = 0.0;
        oracle_hprod(m, c->hy, w, w);
        for (int i = 0; i < m; ++i) y[i] = y[i] + w[i];
    } else {
        oracle_hprod(m, c->hy, y, w);
        for (int i = 0; i < minmn; ++i) w[i] = c->d[i] * w[i];
        for (int i = m; i < n; ++i) w[i] = 0.0;
        oracle_hprod(n, c->hz, w, w);
        for (int i = 0; i < n; ++i) x[i] = x[i] + w[i];
    }
}

/* Allocates d(minmn), hy(m), hz(n), w(max(m,n)).  Returns 0, or 1 when out of memory. */
int oracle_lstp_alloc(oracle_lstp_t *c, int m, int n)
{
    const int minmn = m < n ? m : n, maxmn = m > n ? m : n;
    c->m = m;
    c->n = n;
    c->d = (double *)malloc(sizeof(double) * (size_t)(minmn > 0 ? minmn : 1));
    c->hy = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    c->hz = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    c->w = (double *)malloc(sizeof(double) * (size_t)(maxmn > 0 ? maxmn : 1));
    return (c->d && c->hy && c->hz && c->w) ? 0 : 1;
}

void oracle_lstp_free(oracle_lstp_t *c)
{
    free(c->d); free(c->hy); free(c->hz); free(c->w);
    c->d = c->hy = c->hz = c->w = NULL;
}

/* lstp (:422-505).  x: in = desired solution, out = true solution (changed when m < n and
 * damp > 0).  b(m) out.  acond, rnorm out. */
void oracle_lstp(oracle_lstp_t *c, int nduplc, int npower, double damp, double *x, double *b,
                 double *acond, double *rnorm)
{
    const int m = c->m, n = c->n;
    const int minmn = m < n ? m : n;
    double *d = c->d, *hy = c->hy, *hz = c->hz, *w = c->w;
    const double fourpi = 4.0 * acos(-1.0);                                   /* :436 */
    const double dampsq = damp * damp;
    double alfa = fourpi / m, beta = fourpi / n;                              /* :444-445 */
    for (int i = 1; i <= m; ++i) hy[i - 1] = sin(i * alfa);                   /* :447-449 */
    for (int i = 1; i <= n; ++i) hz[i - 1] = cos(i * beta);                   /* :451-453 */
    alfa = oracle_dnrm2(m, hy, 1);                                            /* :455-458 */
    beta = oracle_dnrm2(n, hz, 1);
    oracle_dscal(m, -1.0 / alfa, hy, 1);
    oracle_dscal(n, -1.0 / beta, hz, 1);
    for (int i = 1; i <= minmn; ++i) {                                        /* :463-468 */
        const int j = (i - 1 + nduplc) / nduplc;
        double t = (double)(j * nduplc);
        t = t / minmn;
        d[i - 1] = __builtin_powi(t, npower);   /* t**npower, integer exponent: repeated multiplication */
    }
    double ac = (d[minmn - 1] * d[minmn - 1] + dampsq) / (d[0] * d[0] + dampsq);  /* :470-471 */
    *acond = sqrt(ac);
    oracle_hprod(n, hz, x, w);                                                /* :478-484 */
    for (int i = m; i < n; ++i) w[i] = 0.0;
    oracle_hprod(n, hz, w, x);
    for (int i = 0; i < minmn; ++i) w[i] = dampsq * w[i] / d[i];              /* :489-491 */
    for (int i = minmn; i < m; ++i) w[i] = 1.0;                               /* :496-498 */
    oracle_hprod(m, hy, w, w);                                                /* :500 */
    *rnorm = oracle_dnrm2(m, w, 1);                                           /* :504-506 */
    oracle_dcopy(m, w, 1, b, 1);
    /* aprod1 uses w as workspace; b already holds r */
    oracle_lstp_aprod(c, 1, m, n, x, b);
}

/* The generated problem in caller arrays: xtrue(n), b(m), d(min(m,n)), hy(m), hz(n). */
int oracle_lstp_generate(int m, int n, int nduplc, int npower, double damp, double *xtrue, double *b, double *d,
                         double *hy, double *hz, double *acond, double *rnorm)
{
    oracle_lstp_t c;
    if (oracle_lstp_alloc(&c, m, n)) return 1;
    for (int j = 1; j <= n; ++j) xtrue[j - 1] = j * 0.1;
    oracle_lstp(&c, nduplc, npower, damp, xtrue, b, acond, rnorm);
    memcpy(d, c.d, sizeof(double) * (size_t)(m < n ? m : n));
    memcpy(hy, c.hy, sizeof(double) * (size_t)m);
    memcpy(hz, c.hz, sizeof(double) * (size_t)n);
    oracle_lstp_free(&c);
    return 0;
}

/* One problem of the suite (:119-272).  x(n), xtrue(n), b(m) are outputs.
 * res[0..15] = acond_lstp, rnorm_lstp, acheck inform, acheck error, istop, itn, anorm, acond,
 *              rnorm, arnorm, xnorm, xcheck inform, test1, test2, test3, enorm.
 * Returns 0, 1 out of memory. */
int oracle_lstp_test(int m, int n, int nduplc, int npower, double damp, double *x, double *xtrue, double *b,
                     double *res)
{
    const double eps = 2.220446049250313e-16;                                 /* epsilon(1.0_wp), :127 */
    oracle_lstp_t c;
    if (oracle_lstp_alloc(&c, m, n)) return 1;
    double *u = (double *)malloc(sizeof(double) * (size_t)m);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    if (!u || !v || !w) return 1;
    for (int j = 1; j <= n; ++j) xtrue[j - 1] = j * 0.1;                      /* :151-154 */
    double acond0, rnorm0;
    oracle_lstp(&c, nduplc, npower, damp, xtrue, b, &acond0, &rnorm0);        /* :175-177 */
    double aerr = 0.0;
    const int ainform = oracle_acheck_op(m, n, oracle_lstp_aprod, &c, eps, &aerr);   /* :184 */
    const double atol = pow(eps, 0.99), btol = atol;                          /* :199-202 */
    const double conlim = 1000.0 * acond0;
    const int itnlim = 4 * (m + n + 50);
    int istop = 0, itn = 0;
    double anorm, acond, rnorm, arnorm, xnorm;
    oracle_lsqr_op(m, n, oracle_lstp_aprod, &c, b, damp, atol, btol, conlim, itnlim, 0, x, NULL, &istop, &itn,
                   &anorm, &acond, &rnorm, &arnorm, &xnorm, NULL, 0);         /* :204-207 */
    double tests[3];
    const int xinform = oracle_xcheck_op(m, n, oracle_lstp_aprod, &c, anorm, damp, eps, b, x, u, v, w, tests);
    for (int j = 0; j < n; ++j) w[j] = x[j] - xtrue[j];                       /* :233-239 */
    const double wnorm = oracle_dnrm2(n, w, 1);
    const double xn = oracle_dnrm2(n, xtrue, 1);
    const double enorm = wnorm / (1.0 + xn);
    res[0] = acond0; res[1] = rnorm0; res[2] = ainform; res[3] = aerr; res[4] = istop; res[5] = itn;
    res[6] = anorm; res[7] = acond; res[8] = rnorm; res[9] = arnorm; res[10] = xnorm; res[11] = xinform;
    res[12] = tests[0]; res[13] = tests[1]; res[14] = tests[2]; res[15] = enorm;
    free(u); free(v); free(w);
    oracle_lstp_free(&c);
    return 0;
}
