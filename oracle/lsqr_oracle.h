/* oracle/lsqr_oracle.h -- TEST INFRASTRUCTURE (see lsqr_oracle.c header). */
#ifndef LSQR_ORACLE_H
#define LSQR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* doubles per per-iteration log record: itn, x(1), rnorm, test1, test2, anorm,
 * acond, phi, dknorm, dxk, alfopt, istop(before the nconv rule), rtol, xnorm */
#define ORACLE_LOG_STRIDE 14

void oracle_dcopy(int n, const double *dx, int incx, double *dy, int incy);
double oracle_ddot(int n, const double *dx, int incx, const double *dy, int incy);
double oracle_dnrm2(int n, const double *x, int incx);
void oracle_dscal(int n, double da, double *dx, int incx);
double oracle_d2norm(double a, double b);

int oracle_aprod(int mode, int m, int n, long long nnz, const int *irow, const int *icol,
                 const double *a, double *x, double *y, double *scratch);
int oracle_validate(int m, int n, long long nnz, const int *irow, const int *icol);

/* the abstract class's operator (src/lsqr.f90:67-82): mode 1 y += A x, mode 2 x += A' y */
typedef void (*oracle_aprod_fn)(void *ctx, int mode, int m, int n, double *x, double *y);

int oracle_lsqr_op(int m, int n, oracle_aprod_fn aprod, void *ctx,
                   const double *b, double damp, double atol, double btol, double conlim,
                   int itnlim, int wantse, double *x, double *se, int *istop, int *itn,
                   double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm,
                   double *log, int logcap);
int oracle_acheck_op(int m, int n, oracle_aprod_fn aprod, void *ctx, double eps, double *err_out);
int oracle_xcheck_op(int m, int n, oracle_aprod_fn aprod, void *ctx, double anorm, double damp, double eps,
                     const double *b, const double *x, double *u, double *v, double *w, double *tests);

int oracle_lsqr_ez(int m, int n, long long nnz, const int *irow, const int *icol, const double *a,
                   const double *b, double damp, double atol, double btol, double conlim,
                   int itnlim, int wantse, double *x, double *se, int *istop, int *itn,
                   double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm,
                   double *log, int logcap);

int oracle_acheck(int m, int n, long long nnz, const int *irow, const int *icol, const double *a,
                  double eps, double *err_out);
int oracle_xcheck(int m, int n, long long nnz, const int *irow, const int *icol, const double *a,
                  double anorm, double damp, double eps, const double *b, const double *x,
                  double *u, double *v, double *w, double *tests);

/* ---- the reference's test-problem class (oracle/lstp_oracle.c; test/lsqrtest_module.f90) ---- */
typedef struct {
    int m, n;
    double *d, *hy, *hz, *w; /* d(min(m,n)), hy(m), hz(n), w(max(m,n)) workspace */
} oracle_lstp_t;
void oracle_lstp_set_sum_order(int order);
void oracle_set_norm_order(int order);
void oracle_set_norm_ulp(int itn, int which, int ulps);   /* test hook: see lsqr_oracle.c */
void oracle_set_accurate_rowsums(int on);                  /* test hook: see lsqr_oracle.c */
void oracle_hprod(int n, const double *hz, const double *x, double *y);
void oracle_lstp_aprod(void *ctx, int mode, int m, int n, double *x, double *y);
int oracle_lstp_alloc(oracle_lstp_t *c, int m, int n);
void oracle_lstp_free(oracle_lstp_t *c);
void oracle_lstp(oracle_lstp_t *c, int nduplc, int npower, double damp, double *x, double *b,
                 double *acond, double *rnorm);
int oracle_lstp_generate(int m, int n, int nduplc, int npower, double damp, double *xtrue, double *b, double *d,
                         double *hy, double *hz, double *acond, double *rnorm);
int oracle_lstp_test(int m, int n, int nduplc, int npower, double damp, double *x, double *xtrue, double *b,
                     double *res);

#ifdef __cplusplus
}
#endif
#endif
