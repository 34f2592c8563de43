// op_api.h -- LSQR on a user-supplied DEVICE operator (included by lsqrhip.hip).
//
// The reference's abstract class `lsqr_solver` (src/lsqr.f90:16-30) leaves `aprod` deferred
// (interface at :67-82: mode 1  y <- y + A x,  mode 2  x <- x + A' y) and runs LSQR (:432-882),
// acheck (:908-994) and xcheck (:1015-1154) on whatever the subclass provides
// (test/lsqrtest_module.f90:35-44 is such a subclass).  Here the operator is a C callback that
// receives DEVICE pointers and a stream and only ENQUEUES work -- the whole iteration stays on
// the device, the host merely feeds the queue a few iterations ahead and polls the stop flag.
//
// The scalar machine and the lazily scaled vectors are exactly those of the EZ path (state.h,
// scalar.h, sequential schedule).  A user operator cannot fold the pending scales into its
// product, so each product is bracketed by one small kernel (k_op_prep):
//
//   mode 1   U <- (-alpha)(U su)        Xs <- V sv      aprod(1, Xs, U)     (:681-682)
//   mode 2   V <- (-beta)(V sv)         Ys <- U su      aprod(2, V, Ys)     (:692-694)
//
// with the rounding sequence of the reference (scale, then multiply by -alpha, then the
// operator adds).  When the product must not happen -- beta == 0 (:691) or the stop flag is up
// -- k_op_prep leaves the target alone and hands the operator a ZERO input vector instead, so
// the enqueued callback adds nothing: no host decision is ever needed.
#pragma once

namespace lsqrhip {

// target <- cy (target sy)   and   scaled <- src sx      (or: target untouched, scaled <- 0)
__global__ __launch_bounds__(VEC_BLOCK) void k_op_prep(double *__restrict__ target, int64_t nt,
                                                       double *__restrict__ scaled, const double *__restrict__ src,
                                                       int64_t ns, const SpmvCoef *__restrict__ coef,
                                                       const int *__restrict__ stop)
{
    const bool idle = *stop != 0 || coef->skip != 0;
    const double sx = coef->sx, sy = coef->sy, cy = coef->cy;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    const int64_t i0 = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x;
    if (idle) {
        for (int64_t i = i0; i < ns; i += stride) scaled[i] = 0.0;
        return;
    }
    for (int64_t i = i0; i < nt; i += stride) target[i] = cy * (target[i] * sy);
    for (int64_t i = i0; i < ns; i += stride) scaled[i] = src[i] * sx;
}

}  // namespace lsqrhip

static int op_call(H *h, int mode, double *d_x, double *d_y)
{
    const int rc = h->op(h->op_user, mode, h->m, h->n, d_x, d_y, (void *)h->stream);
    if (rc != 0) return fail(LSQRHIP_ERR_ARG, "user aprod returned " + std::to_string(rc));
    return LSQRHIP_OK;
}

// One LSQR iteration on the operator, sequential schedule (src/lsqr.f90:673-852).
static int op_iteration(H *h)
{
    LsqrState *st = h->d_state;
    hipStream_t s = h->stream;
    const int64_t m = h->m, n = h->n;
    const int gmn = vec_grid(std::max(m, n));
    hipLaunchKernelGGL(k_op_prep, dim3(gmn), dim3(VEC_BLOCK), 0, s, h->U, m, h->opX, (const double *)h->V, n,
                       (const SpmvCoef *)&st->c1, (const int *)&st->stop);
    RET(op_call(h, 1, h->opX, h->U));
    hipLaunchKernelGGL(k_sumsq3, dim3(h->vgrid_m), dim3(VEC_BLOCK), 0, s, (const double *)h->U, m, h->partials);
    hipLaunchKernelGGL((k_s1<true, true>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_m,
                       (const double *)nullptr, st);
    hipLaunchKernelGGL(k_op_prep, dim3(gmn), dim3(VEC_BLOCK), 0, s, h->V, n, h->opY, (const double *)h->U, m,
                       (const SpmvCoef *)&st->c2, (const int *)&st->stop);
    RET(op_call(h, 2, h->V, h->opY));
    hipLaunchKernelGGL(k_sumsq3, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, (const double *)h->V, n, h->partials);
    hipLaunchKernelGGL((k_s2<true, true>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                       (const double *)nullptr, st);
    launch_update(h, h->partials, nullptr, nullptr);
    hipLaunchKernelGGL(k_s3<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                       (const double *)nullptr, st, (const void *)h->X, 0, h->d_log);
    HIPCHK(hipGetLastError());
    return LSQRHIP_OK;
}

// solve_core for operator handles: same state machine, same outputs.
static int solve_op(H *h, const double *b, bool b_on_device, double damp, double atol, double btol, double conlim,
                    int itnlim, int wantse, int want_log, double *x, double *se, bool out_on_device, int *istop,
                    int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    const auto t_host0 = std::chrono::steady_clock::now();
    hipStream_t s = h->stream;
    const int m = h->m, n = h->n;
    LsqrState *st = h->d_state;
    RET(prepare_log(h, itnlim, want_log));
    RET(upload_initial_state(h, damp, atol, btol, conlim, itnlim, wantse, want_log));
    HIPCHK(hipMemcpyAsync(h->d_state, h->h_state, sizeof(LsqrState), hipMemcpyHostToDevice, s));
    if (m > 0)
        HIPCHK(hipMemcpyAsync(h->U, b, sizeof(double) * (size_t)m,
                              b_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (n > 0) {
        HIPCHK(hipMemsetAsync(h->V, 0, sizeof(double) * (size_t)n, s));
        HIPCHK(hipMemsetAsync(h->X, 0, sizeof(double) * (size_t)n, s));
        HIPCHK(hipMemsetAsync(h->W, 0, sizeof(double) * (size_t)n, s));
        if (wantse) HIPCHK(hipMemsetAsync(h->SE, 0, sizeof(double) * (size_t)n, s));
    }
    // beta = norm(u); u /= beta; v = A'u; alpha = norm(v); v /= alpha; w = v      (:632-644)
    hipLaunchKernelGGL(k_sumsq3, dim3(h->vgrid_m), dim3(VEC_BLOCK), 0, s, (const double *)h->U, (int64_t)m,
                       h->partials);
    hipLaunchKernelGGL(k_s_init1<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_m,
                       (const double *)nullptr, st, (NormSlot *)nullptr);
    hipLaunchKernelGGL(k_op_prep, dim3(vec_grid(std::max(m, n))), dim3(VEC_BLOCK), 0, s, h->V, (int64_t)n, h->opY,
                       (const double *)h->U, (int64_t)m, (const SpmvCoef *)&st->c2, (const int *)h->d_zero);
    RET(op_call(h, 2, h->V, h->opY));
    hipLaunchKernelGGL(k_sumsq3, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, (const double *)h->V, (int64_t)n,
                       h->partials);
    hipLaunchKernelGGL((k_s_init2<true, true>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                       (const double *)nullptr, st);
    hipLaunchKernelGGL(k_copy_scale, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->W, (const double *)h->V, (int64_t)n,
                       (const LsqrState *)st);
    HIPCHK(hipGetLastError());

    lsqrhip_timing_t &tm = h->timing;
    tm = lsqrhip_timing_t{};
    tm.vec_bytes = 40ll * n + (wantse ? 16ll * n : 0);
    // Iterations enqueued ahead of each poll: the callbacks past the stopping iteration still
    // run (on zero vectors), so keep the batch short.
    const int G = std::max(1, h->op_batch);
    h->loop_bracketed = true;
    HIPCHK(hipEventRecord(h->ev_loop0, s));
    const int64_t max_batches = (int64_t)std::max(itnlim, 0) / G + 2;
    for (int64_t batch = 0;; ++batch) {
        if (batch > max_batches)
            return fail(LSQRHIP_ERR_HIP, "iteration loop did not terminate (device state not advancing)");
        for (int k = 0; k < G; ++k) RET(op_iteration(h));
        HIPCHK(hipMemcpyAsync(h->h_state, st, sizeof(LsqrState), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (h->h_state->stop != 0) break;
    }
    HIPCHK(hipEventRecord(h->ev_loop1, s));
    return finish_solve(h, wantse, want_log, x, se, out_on_device, istop, itn, anorm, acond, rnorm, arnorm, xnorm,
                        false, t_host0);
}

extern "C" int lsqrhip_create_operator(int m, int n, lsqrhip_aprod_fn aprod, void *user, lsqrhip_handle_t *out)
{
    if (!aprod) return fail(LSQRHIP_ERR_ARG, "null aprod callback");
    H *h = nullptr;
    RET(new_handle(m, n, 0, &h));
    h->op = aprod;
    h->op_user = user;
    int rc = alloc_workspace(h);
    if (rc == LSQRHIP_OK) {
        hipError_t e1 = hipMalloc((void **)&h->opX, sizeof(double) * (size_t)std::max(n, 1));
        hipError_t e2 = hipMalloc((void **)&h->opY, sizeof(double) * (size_t)std::max(m, 1));
        if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(LSQRHIP_ERR_ALLOC, "operator scratch vectors");
    }
    if (rc != LSQRHIP_OK) {
        std::string keep = g_last_error;
        lsqrhip_destroy(h);
        g_last_error = keep;
        return rc;
    }
    *out = h;
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// The reference's test operator  A = HY * D * HZ  on the device
// (test/lsqrtest_module.f90: hprod :385-403, aprod1 :319-343, aprod2 :353-377, lstp :422-505)
// ---------------------------------------------------------------------------
namespace lsqrhip {

// out_i = (x_i - (2 s) h_i) [* d_i for i < nd],  s = sum of `partials` (hprod, then the diagonal);
// i in [nx, nout) is set to zero (the `w(i) = zero` loops of aprod1 / aprod2).
__global__ __launch_bounds__(VEC_BLOCK) void k_hprod_scale(double *__restrict__ out, int64_t nout,
                                                           const double *__restrict__ x, const double *__restrict__ hv,
                                                           int64_t nx, const double *__restrict__ d, int64_t nd,
                                                           const double *__restrict__ partials, int np)
{
    __shared__ double red[VEC_BLOCK / WAVE + 1];
    double s = block_sum_all<VEC_BLOCK>(partials, np, red);
    s = s + s;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < nout; i += stride) {
        double v = 0.0;
        if (i < nx) {
            v = x[i] - s * hv[i];
            if (i < nd) v = d[i] * v;
        }
        out[i] = v;
    }
}

// y_i <- y_i + (w_i - (2 s) h_i)        (second hprod of aprod1 / aprod2, then the add)
__global__ __launch_bounds__(VEC_BLOCK) void k_hprod_add(double *__restrict__ y, const double *__restrict__ w,
                                                         const double *__restrict__ hv, int64_t n,
                                                         const double *__restrict__ partials, int np)
{
    __shared__ double red[VEC_BLOCK / WAVE + 1];
    double s = block_sum_all<VEC_BLOCK>(partials, np, red);
    s = s + s;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride)
        y[i] = y[i] + (w[i] - s * hv[i]);
}

}  // namespace lsqrhip

struct LstpOp {
    int m = 0, n = 0, minmn = 0;
    double *d = nullptr, *hy = nullptr, *hz = nullptr, *w = nullptr, *part = nullptr;
    double *b = nullptr, *xtrue = nullptr;  // device copies of the generated right-hand side / true solution
    std::vector<double> h_d, h_hy, h_hz, h_b, h_xtrue;
};

static int lstp_aprod(void *user, int mode, int m, int n, double *x, double *y, void *stream)
{
    LstpOp *o = (LstpOp *)user;
    hipStream_t s = (hipStream_t)stream;
    // mode 1: w = D HZ x (n -> m entries), y += HY w.   mode 2: w = D' HY y (m -> n), x += HZ w.
    const int64_t nin = mode == 1 ? n : m, nout = mode == 1 ? m : n;
    const double *hin = mode == 1 ? o->hz : o->hy, *hout = mode == 1 ? o->hy : o->hz;
    const double *vin = mode == 1 ? x : y;
    double *vout = mode == 1 ? y : x;
    const int gin = vec_grid(nin), gout = vec_grid(nout);
    hipLaunchKernelGGL(k_dot, dim3(gin), dim3(VEC_BLOCK), 0, s, hin, vin, nin, o->part);
    hipLaunchKernelGGL(k_hprod_scale, dim3(gout), dim3(VEC_BLOCK), 0, s, o->w, nout, vin, hin, nin, (const double *)o->d,
                       (int64_t)o->minmn, (const double *)o->part, gin);
    hipLaunchKernelGGL(k_dot, dim3(gout), dim3(VEC_BLOCK), 0, s, hout, (const double *)o->w, nout, o->part);
    hipLaunchKernelGGL(k_hprod_add, dim3(gout), dim3(VEC_BLOCK), 0, s, vout, (const double *)o->w, hout, nout,
                       (const double *)o->part, gout);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

static void lstp_free(void *user)
{
    LstpOp *o = (LstpOp *)user;
    for (double *p : {o->d, o->hy, o->hz, o->w, o->part, o->b, o->xtrue})
        if (p) (void)hipFree(p);
    delete o;
}

// dnrm2 as the reference computes it (src/lsqrblas.f90:123-159: scaled sum of squares)
static double host_dnrm2(const std::vector<double> &x)
{
    const size_t n = x.size();
    if (n < 1) return 0.0;
    if (n == 1) return std::fabs(x[0]);
    double scale = 0.0, ssq = 1.0;
    for (size_t i = 0; i < n; ++i) {
        if (x[i] != 0.0) {
            const double absxi = std::fabs(x[i]);
            if (scale < absxi) {
                ssq = 1.0 + ssq * (scale / absxi) * (scale / absxi);
                scale = absxi;
            } else {
                ssq = ssq + (absxi / scale) * (absxi / scale);
            }
        }
    }
    return scale * std::sqrt(ssq);
}

static void host_hprod(const std::vector<double> &hz, size_t n, const double *x, double *y)
{
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s = hz[i] * x[i] + s;
    s = s + s;
    for (size_t i = 0; i < n; ++i) y[i] = x[i] - s * hz[i];
}

// lstp (test/lsqrtest_module.f90:422-505), generated on the host in the reference's own
// operation order (O(m + n) work, once), then uploaded.
extern "C" int lsqrhip_lstp_create(int m, int n, int nduplc, int npower, double damp, lsqrhip_handle_t *out,
                                   double *acond_out, double *rnorm_out)
{
    if (m < 1 || n < 1 || nduplc < 1) return fail(LSQRHIP_ERR_ARG, "lstp needs m, n, nduplc >= 1");
    LstpOp *o = new LstpOp();
    o->m = m;
    o->n = n;
    const int minmn = std::min(m, n), maxmn = std::max(m, n);
    o->minmn = minmn;
    std::vector<double> &d = o->h_d, &hy = o->h_hy, &hz = o->h_hz, &b = o->h_b, &x = o->h_xtrue;
    d.resize(minmn); hy.resize(m); hz.resize(n); b.resize(m); x.resize(n);
    std::vector<double> w(maxmn);
    for (int j = 1; j <= n; ++j) x[j - 1] = j * 0.1;                            // test :151-154
    const double fourpi = 4.0 * std::acos(-1.0);                                // :436
    const double dampsq = damp * damp;
    double alfa = fourpi / m, beta = fourpi / n;
    for (int i = 1; i <= m; ++i) hy[i - 1] = std::sin(i * alfa);
    for (int i = 1; i <= n; ++i) hz[i - 1] = std::cos(i * beta);
    alfa = host_dnrm2(hy);
    beta = host_dnrm2(hz);
    for (double &v : hy) v = (-1.0 / alfa) * v;
    for (double &v : hz) v = (-1.0 / beta) * v;
    for (int i = 1; i <= minmn; ++i) {                                          // :463-468
        const int j = (i - 1 + nduplc) / nduplc;
        double t = (double)(j * nduplc);
        t = t / minmn;
        d[i - 1] = __builtin_powi(t, npower);
    }
    const double acond = std::sqrt((d[minmn - 1] * d[minmn - 1] + dampsq) / (d[0] * d[0] + dampsq));
    host_hprod(hz, n, x.data(), w.data());                                      // :478-484
    for (int i = m; i < n; ++i) w[i] = 0.0;
    {
        std::vector<double> t(w.begin(), w.begin() + n);
        host_hprod(hz, n, t.data(), x.data());
    }
    for (int i = 0; i < minmn; ++i) w[i] = dampsq * w[i] / d[i];                // :489-491
    for (int i = minmn; i < m; ++i) w[i] = 1.0;                                 // :496-498
    {
        std::vector<double> t(w.begin(), w.begin() + m);
        host_hprod(hy, m, t.data(), w.data());                                  // :500
    }
    const double rnorm = host_dnrm2(std::vector<double>(w.begin(), w.begin() + m));
    for (int i = 0; i < m; ++i) b[i] = w[i];
    {   // b = r + A x   (aprod1, :319-343)
        std::vector<double> t(maxmn), t2(maxmn);
        host_hprod(hz, n, x.data(), t.data());
        for (int i = 0; i < minmn; ++i) t[i] = d[i] * t[i];
        for (int i = n; i < m; ++i) t[i] = 0.0;
        host_hprod(hy, m, t.data(), t2.data());
        for (int i = 0; i < m; ++i) b[i] = b[i] + t2[i];
    }
    if (acond_out) *acond_out = acond;
    if (rnorm_out) *rnorm_out = rnorm;

    int rc = use_device();
    if (rc != LSQRHIP_OK) {
        delete o;
        return rc;
    }
    auto up = [&](double **p, const std::vector<double> &v, size_t cap) -> bool {
        if (hipMalloc((void **)p, sizeof(double) * std::max<size_t>(cap, 1)) != hipSuccess) return false;
        return v.empty() || hipMemcpy(*p, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice) == hipSuccess;
    };
    const bool ok = up(&o->d, d, d.size()) && up(&o->hy, hy, hy.size()) && up(&o->hz, hz, hz.size()) &&
                    up(&o->b, b, b.size()) && up(&o->xtrue, x, x.size()) && up(&o->w, {}, (size_t)maxmn) &&
                    up(&o->part, {}, (size_t)VEC_MAX_GRID);
    if (!ok) {
        lstp_free(o);
        return fail(LSQRHIP_ERR_ALLOC, "lstp device vectors");
    }
    rc = lsqrhip_create_operator(m, n, lstp_aprod, o, out);
    if (rc != LSQRHIP_OK) {
        lstp_free(o);
        return rc;
    }
    (*out)->op_free = lstp_free;
    (*out)->op_is_lstp = true;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_lstp_vectors(lsqrhip_handle_t h, double *xtrue, double *b, double *d, double *hy, double *hz,
                                    const double **d_b)
{
    if (!h || !h->op_is_lstp) return fail(LSQRHIP_ERR_ARG, "not a handle of lsqrhip_lstp_create");
    const LstpOp *o = (const LstpOp *)h->op_user;
    if (xtrue) std::memcpy(xtrue, o->h_xtrue.data(), sizeof(double) * o->h_xtrue.size());
    if (b) std::memcpy(b, o->h_b.data(), sizeof(double) * o->h_b.size());
    if (d) std::memcpy(d, o->h_d.data(), sizeof(double) * o->h_d.size());
    if (hy) std::memcpy(hy, o->h_hy.data(), sizeof(double) * o->h_hy.size());
    if (hz) std::memcpy(hz, o->h_hz.data(), sizeof(double) * o->h_hz.size());
    if (d_b) *d_b = o->b;
    return LSQRHIP_OK;
}
