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
// (VT = float: the vectors of a REAL32 operator handle -- lsqrhip_create_operator_f32; the arithmetic stays binary64)
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_op_prep(VT *__restrict__ target, int64_t nt,
                                                       VT *__restrict__ scaled, const VT *__restrict__ src,
                                                       int64_t ns, const SpmvCoef *__restrict__ coef,
                                                       const int *__restrict__ stop)
{
    const bool idle = *stop != 0 || coef->skip != 0;
    const double sx = coef->sx, sy = coef->sy, cy = coef->cy;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    const int64_t i0 = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x;
    if (idle) {
        for (int64_t i = i0; i < ns; i += stride) scaled[i] = (VT)0;
        return;
    }
    for (int64_t i = i0; i < nt; i += stride) target[i] = (VT)(cy * ((double)target[i] * sy));
    for (int64_t i = i0; i < ns; i += stride) scaled[i] = (VT)((double)src[i] * sx);
}

}  // namespace lsqrhip

static int op_call(H *h, int mode, double *d_x, double *d_y)
{
    const int rc = h->op(h->op_user, mode, h->m, h->n, d_x, d_y, (void *)h->stream);
    if (rc != 0) return fail(LSQRHIP_ERR_ARG, "user aprod returned " + std::to_string(rc));
    return LSQRHIP_OK;
}

// One LSQR iteration on the operator, sequential schedule (src/lsqr.f90:673-852).
template <typename VT>
static int op_iteration_T(H *h)
{
    LsqrState *st = h->d_state;
    hipStream_t s = h->stream;
    const int64_t m = h->m, n = h->n;
    const int gmn = vec_grid(std::max(m, n));
    VT *U = reinterpret_cast<VT *>(h->U), *V = reinterpret_cast<VT *>(h->V);
    VT *opX = reinterpret_cast<VT *>(h->opX), *opY = reinterpret_cast<VT *>(h->opY);
    hipLaunchKernelGGL(k_op_prep<VT>, dim3(gmn), dim3(VEC_BLOCK), 0, s, U, m, opX, (const VT *)V, n,
                       (const SpmvCoef *)&st->c1, (const int *)&st->stop);
    RET(op_call(h, 1, h->opX, h->U));
    hipLaunchKernelGGL(k_sumsq3<VT>, dim3(h->vgrid_m), dim3(VEC_BLOCK), 0, s, (const VT *)U, m, h->partials);
    hipLaunchKernelGGL((k_s1<true, true>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_m,
                       (const double *)nullptr, st);
    hipLaunchKernelGGL(k_op_prep<VT>, dim3(gmn), dim3(VEC_BLOCK), 0, s, V, n, opY, (const VT *)U, m,
                       (const SpmvCoef *)&st->c2, (const int *)&st->stop);
    RET(op_call(h, 2, h->V, h->opY));
    hipLaunchKernelGGL(k_sumsq3<VT>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, (const VT *)V, n, h->partials);
    hipLaunchKernelGGL((k_s2<true, true>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                       (const double *)nullptr, st);
    launch_update(h, h->partials, nullptr, nullptr);
    hipLaunchKernelGGL(k_s3<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                       (const double *)nullptr, st, (const void *)h->X, h->f32 ? 1 : 0, h->d_log);
    HIPCHK(hipGetLastError());
    return LSQRHIP_OK;
}
static int op_iteration(H *h) { return h->f32 ? op_iteration_T<float>(h) : op_iteration_T<double>(h); }

// the start of a solve on the operator: beta = norm(u); u /= beta; v = A'u; alpha = norm(v); v /= alpha; w = v (:632-644)
template <typename VT>
static int op_start_T(H *h)
{
    hipStream_t s = h->stream;
    const int m = h->m, n = h->n;
    LsqrState *st = h->d_state;
    VT *U = reinterpret_cast<VT *>(h->U), *V = reinterpret_cast<VT *>(h->V), *W = reinterpret_cast<VT *>(h->W);
    hipLaunchKernelGGL(k_sumsq3<VT>, dim3(h->vgrid_m), dim3(VEC_BLOCK), 0, s, (const VT *)U, (int64_t)m, h->partials);
    hipLaunchKernelGGL(k_s_init1<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_m,
                       (const double *)nullptr, st, (NormSlot *)nullptr);
    hipLaunchKernelGGL(k_op_prep<VT>, dim3(vec_grid(std::max(m, n))), dim3(VEC_BLOCK), 0, s, V, (int64_t)n,
                       reinterpret_cast<VT *>(h->opY), (const VT *)U, (int64_t)m, (const SpmvCoef *)&st->c2,
                       (const int *)h->d_zero);
    RET(op_call(h, 2, h->V, h->opY));
    hipLaunchKernelGGL(k_sumsq3<VT>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, (const VT *)V, (int64_t)n, h->partials);
    hipLaunchKernelGGL((k_s_init2<true, true>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                       (const double *)nullptr, st);
    hipLaunchKernelGGL(k_copy_scale<VT>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, W, (const VT *)V, (int64_t)n,
                       (const LsqrState *)st);
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
    const size_t esz = h->f32 ? sizeof(float) : sizeof(double);   // (b, x, se: float arrays for a REAL32 operator handle)
    if (m > 0)
        HIPCHK(hipMemcpyAsync(h->U, b, esz * (size_t)m,
                              b_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (n > 0) {
        HIPCHK(hipMemsetAsync(h->V, 0, esz * (size_t)n, s));
        HIPCHK(hipMemsetAsync(h->X, 0, esz * (size_t)n, s));
        HIPCHK(hipMemsetAsync(h->W, 0, esz * (size_t)n, s));
        if (wantse) HIPCHK(hipMemsetAsync(h->SE, 0, esz * (size_t)n, s));
    }
    RET(h->f32 ? op_start_T<float>(h) : op_start_T<double>(h));

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

static int create_operator_any(int m, int n, lsqrhip_aprod_fn aprod, void *user, bool f32, lsqrhip_handle_t *out)
{
    if (!aprod) return fail(LSQRHIP_ERR_ARG, "null aprod callback");
    H *h = nullptr;
    RET(new_handle(m, n, 0, &h));
    h->op = aprod;
    h->op_user = user;
    h->f32 = h->io32 = f32;
    const size_t esz = f32 ? sizeof(float) : sizeof(double);
    int rc = alloc_workspace(h);
    if (rc == LSQRHIP_OK) {
        hipError_t e1 = hipMalloc((void **)&h->opX, esz * (size_t)std::max(n, 1));
        hipError_t e2 = hipMalloc((void **)&h->opY, esz * (size_t)std::max(m, 1));
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

extern "C" int lsqrhip_create_operator(int m, int n, lsqrhip_aprod_fn aprod, void *user, lsqrhip_handle_t *out)
{
    return create_operator_any(m, n, aprod, user, false, out);
}

// The REAL32 build of the abstract class (src/lsqr_kinds.F90:16-17 applies to lsqr_solver too, src/lsqr.f90:16-30):
// the operator's vectors -- and every work vector of the iteration -- are real32 arrays on the device; the callback
// receives float pointers (in the same argument slots); solve with lsqrhip_solve_f32 / lsqrhip_solve_device_f32.
extern "C" int lsqrhip_create_operator_f32(int m, int n, lsqrhip_aprod_f32_fn aprod, void *user, lsqrhip_handle_t *out)
{
    return create_operator_any(m, n, reinterpret_cast<lsqrhip_aprod_fn>(aprod), user, true, out);
}

// ---------------------------------------------------------------------------
// The reference's test operator  A = HY * D * HZ  on the device
// (test/lsqrtest_module.f90: hprod :385-403, aprod1 :319-343, aprod2 :353-377, lstp :422-505)
// ---------------------------------------------------------------------------
namespace lsqrhip {

// partials[b] = sum over this workgroup's share of x[i] * y[i] (vec.h k_dot for either storage type)
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_dot_T(const VT *__restrict__ x, const VT *__restrict__ y, int64_t n,
                                                     double *__restrict__ partials)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) s += (double)x[i] * (double)y[i];
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// out_i = (x_i - (2 s) h_i) [* d_i for i < nd],  s = sum of `partials` (hprod, then the diagonal);
// i in [nx, nout) is set to zero (the `w(i) = zero` loops of aprod1 / aprod2).
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_hprod_scale(VT *__restrict__ out, int64_t nout,
                                                           const VT *__restrict__ x, const VT *__restrict__ hv,
                                                           int64_t nx, const VT *__restrict__ d, int64_t nd,
                                                           const double *__restrict__ partials, int np)
{
    __shared__ double red[VEC_BLOCK / WAVE + 1];
    double s = block_sum_all<VEC_BLOCK>(partials, np, red);
    s = s + s;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < nout; i += stride) {
        double v = 0.0;
        if (i < nx) {
            v = (double)x[i] - s * (double)hv[i];
            if (i < nd) v = (double)d[i] * v;
        }
        out[i] = (VT)v;
    }
}

// y_i <- y_i + (w_i - (2 s) h_i)        (second hprod of aprod1 / aprod2, then the add)
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_hprod_add(VT *__restrict__ y, const VT *__restrict__ w,
                                                         const VT *__restrict__ hv, int64_t n,
                                                         const double *__restrict__ partials, int np)
{
    __shared__ double red[VEC_BLOCK / WAVE + 1];
    double s = block_sum_all<VEC_BLOCK>(partials, np, red);
    s = s + s;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride)
        y[i] = (VT)((double)y[i] + ((double)w[i] - s * (double)hv[i]));
}

}  // namespace lsqrhip

// (device arrays of T = double or, for the REAL32 build, float; the host copies keep the generated values as doubles)
struct LstpOp {
    int m = 0, n = 0, minmn = 0;
    bool f32 = false;
    void *d = nullptr, *hy = nullptr, *hz = nullptr, *w = nullptr, *b = nullptr, *xtrue = nullptr;
    double *part = nullptr;
    std::vector<double> h_d, h_hy, h_hz, h_b, h_xtrue;
};

template <typename VT>
static int lstp_aprod_T(LstpOp *o, int mode, int m, int n, VT *x, VT *y, hipStream_t s)
{
    // mode 1: w = D HZ x (n -> m entries), y += HY w.   mode 2: w = D' HY y (m -> n), x += HZ w.
    const int64_t nin = mode == 1 ? n : m, nout = mode == 1 ? m : n;
    const VT *hin = (const VT *)(mode == 1 ? o->hz : o->hy), *hout = (const VT *)(mode == 1 ? o->hy : o->hz);
    const VT *vin = mode == 1 ? x : y;
    VT *vout = mode == 1 ? y : x;
    VT *w = (VT *)o->w;
    const int gin = vec_grid(nin), gout = vec_grid(nout);
    if (std::is_same<VT, double>::value)   // (binary64: vec.h's k_dot, whose partial sums the goldens of round 2 hold)
        hipLaunchKernelGGL(k_dot, dim3(gin), dim3(VEC_BLOCK), 0, s, (const double *)hin, (const double *)vin, nin, o->part);
    else
        hipLaunchKernelGGL(k_dot_T<VT>, dim3(gin), dim3(VEC_BLOCK), 0, s, hin, vin, nin, o->part);
    hipLaunchKernelGGL(k_hprod_scale<VT>, dim3(gout), dim3(VEC_BLOCK), 0, s, w, nout, vin, hin, nin, (const VT *)o->d,
                       (int64_t)o->minmn, (const double *)o->part, gin);
    if (std::is_same<VT, double>::value)
        hipLaunchKernelGGL(k_dot, dim3(gout), dim3(VEC_BLOCK), 0, s, (const double *)hout, (const double *)w, nout, o->part);
    else
        hipLaunchKernelGGL(k_dot_T<VT>, dim3(gout), dim3(VEC_BLOCK), 0, s, hout, (const VT *)w, nout, o->part);
    hipLaunchKernelGGL(k_hprod_add<VT>, dim3(gout), dim3(VEC_BLOCK), 0, s, vout, (const VT *)w, hout, nout,
                       (const double *)o->part, gout);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

static int lstp_aprod(void *user, int mode, int m, int n, double *x, double *y, void *stream)
{
    LstpOp *o = (LstpOp *)user;
    if (o->f32) return lstp_aprod_T<float>(o, mode, m, n, reinterpret_cast<float *>(x), reinterpret_cast<float *>(y), (hipStream_t)stream);
    return lstp_aprod_T<double>(o, mode, m, n, x, y, (hipStream_t)stream);
}

static void lstp_free(void *user)
{
    LstpOp *o = (LstpOp *)user;
    for (void *p : {o->d, o->hy, o->hz, o->w, (void *)o->part, o->b, o->xtrue})
        if (p) (void)hipFree(p);
    delete o;
}

// dnrm2 as the reference computes it (src/lsqrblas.f90:123-159: scaled sum of squares), in the working precision T
template <typename T>
static T host_dnrm2(const std::vector<T> &x, size_t n)
{
    if (n < 1) return (T)0;
    if (n == 1) return std::fabs(x[0]);
    T scale = 0, ssq = 1;
    for (size_t i = 0; i < n; ++i) {
        if (x[i] != (T)0) {
            const T absxi = std::fabs(x[i]);
            if (scale < absxi) {
                ssq = (T)1 + ssq * (scale / absxi) * (scale / absxi);
                scale = absxi;
            } else {
                ssq = ssq + (absxi / scale) * (absxi / scale);
            }
        }
    }
    return scale * std::sqrt(ssq);
}

template <typename T>
static void host_hprod(const std::vector<T> &hz, size_t n, const T *x, T *y)
{
    T s = 0;
    for (size_t i = 0; i < n; ++i) s = hz[i] * x[i] + s;
    s = s + s;
    for (size_t i = 0; i < n; ++i) y[i] = x[i] - s * hz[i];
}

// lstp (test/lsqrtest_module.f90:422-505), generated on the host in the reference's own operation order AND working
// precision T (O(m + n) work, once), then uploaded as T.
template <typename T>
static int lstp_create_T(int m, int n, int nduplc, int npower, double damp_, lsqrhip_handle_t *out, double *acond_out,
                         double *rnorm_out)
{
    if (m < 1 || n < 1 || nduplc < 1) return fail(LSQRHIP_ERR_ARG, "lstp needs m, n, nduplc >= 1");
    LstpOp *o = new LstpOp();
    o->m = m;
    o->n = n;
    o->f32 = std::is_same<T, float>::value;
    const int minmn = std::min(m, n), maxmn = std::max(m, n);
    o->minmn = minmn;
    const T damp = (T)damp_;
    std::vector<T> d(minmn), hy(m), hz(n), b(m), x(n), w(maxmn);
    for (int j = 1; j <= n; ++j) x[j - 1] = (T)j * (T)0.1;                      // test :151-154
    const T fourpi = (T)4 * std::acos((T)-1);                                   // :436
    const T dampsq = damp * damp;
    T alfa = fourpi / (T)m, beta = fourpi / (T)n;
    for (int i = 1; i <= m; ++i) hy[i - 1] = std::sin((T)i * alfa);
    for (int i = 1; i <= n; ++i) hz[i - 1] = std::cos((T)i * beta);
    alfa = host_dnrm2(hy, (size_t)m);
    beta = host_dnrm2(hz, (size_t)n);
    for (T &v : hy) v = ((T)-1 / alfa) * v;
    for (T &v : hz) v = ((T)-1 / beta) * v;
    for (int i = 1; i <= minmn; ++i) {                                          // :463-468
        const int j = (i - 1 + nduplc) / nduplc;
        T t = (T)(j * nduplc);
        t = t / (T)minmn;
        T pw = 1;
        for (int k = 0; k < npower; ++k) pw = pw * t;
        d[i - 1] = std::is_same<T, double>::value ? (T)__builtin_powi((double)t, npower) : pw;
    }
    const T acond = std::sqrt((d[minmn - 1] * d[minmn - 1] + dampsq) / (d[0] * d[0] + dampsq));
    host_hprod(hz, (size_t)n, x.data(), w.data());                              // :478-484
    for (int i = m; i < n; ++i) w[i] = 0;
    {
        std::vector<T> t(w.begin(), w.begin() + n);
        host_hprod(hz, (size_t)n, t.data(), x.data());
    }
    for (int i = 0; i < minmn; ++i) w[i] = dampsq * w[i] / d[i];                // :489-491
    for (int i = minmn; i < m; ++i) w[i] = 1;                                   // :496-498
    {
        std::vector<T> t(w.begin(), w.begin() + m);
        host_hprod(hy, (size_t)m, t.data(), w.data());                          // :500
    }
    const T rnorm = host_dnrm2(w, (size_t)m);
    for (int i = 0; i < m; ++i) b[i] = w[i];
    {   // b = r + A x   (aprod1, :319-343)
        std::vector<T> t(maxmn), t2(maxmn);
        host_hprod(hz, (size_t)n, x.data(), t.data());
        for (int i = 0; i < minmn; ++i) t[i] = d[i] * t[i];
        for (int i = n; i < m; ++i) t[i] = 0;
        host_hprod(hy, (size_t)m, t.data(), t2.data());
        for (int i = 0; i < m; ++i) b[i] = b[i] + t2[i];
    }
    if (acond_out) *acond_out = (double)acond;
    if (rnorm_out) *rnorm_out = (double)rnorm;
    o->h_d.assign(d.begin(), d.end());
    o->h_hy.assign(hy.begin(), hy.end());
    o->h_hz.assign(hz.begin(), hz.end());
    o->h_b.assign(b.begin(), b.end());
    o->h_xtrue.assign(x.begin(), x.end());

    int rc = use_device();
    if (rc != LSQRHIP_OK) {
        delete o;
        return rc;
    }
    auto up = [&](void **p, const std::vector<T> &v, size_t cap) -> bool {
        if (hipMalloc(p, sizeof(T) * std::max<size_t>(cap, 1)) != hipSuccess) return false;
        return v.empty() || hipMemcpy(*p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) == hipSuccess;
    };
    const bool ok = up(&o->d, d, d.size()) && up(&o->hy, hy, hy.size()) && up(&o->hz, hz, hz.size()) &&
                    up(&o->b, b, b.size()) && up(&o->xtrue, x, x.size()) && up(&o->w, {}, (size_t)maxmn) &&
                    hipMalloc((void **)&o->part, sizeof(double) * VEC_MAX_GRID) == hipSuccess;
    if (!ok) {
        lstp_free(o);
        return fail(LSQRHIP_ERR_ALLOC, "lstp device vectors");
    }
    rc = create_operator_any(m, n, lstp_aprod, o, o->f32, out);
    if (rc != LSQRHIP_OK) {
        lstp_free(o);
        return rc;
    }
    (*out)->op_free = lstp_free;
    (*out)->op_is_lstp = true;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_lstp_create(int m, int n, int nduplc, int npower, double damp, lsqrhip_handle_t *out,
                                   double *acond_out, double *rnorm_out)
{
    return lstp_create_T<double>(m, n, nduplc, npower, damp, out, acond_out, rnorm_out);
}

// ... in the reference's REAL32 build (src/lsqr_kinds.F90:16-17: wp = real32 in the test module too): the problem is
// generated in binary32 arithmetic and lives in real32 arrays on the device (a handle of lsqrhip_create_operator_f32)
extern "C" int lsqrhip_lstp_create_f32(int m, int n, int nduplc, int npower, double damp, lsqrhip_handle_t *out,
                                       double *acond_out, double *rnorm_out)
{
    return lstp_create_T<float>(m, n, nduplc, npower, damp, out, acond_out, rnorm_out);
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
    if (d_b) *d_b = (const double *)o->b;   // (a float array for a handle of lsqrhip_lstp_create_f32)
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// acheck / xcheck for REAL32 handles -- matrix handles of lsqrhip_create_f32 and operator handles of
// lsqrhip_create_operator_f32 alike (src/lsqr.f90:908-994, 1015-1154 with wp = real32: real32 vectors, here
// with binary64 arithmetic between them, like everything else of the REAL32 build)
// ---------------------------------------------------------------------------
namespace lsqrhip {

__global__ __launch_bounds__(VEC_BLOCK) void k_f32_acheck_fill(float *__restrict__ x, int64_t n, int inverse)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) {
        const double s = sqrt((double)(i + 2));
        x[i] = (float)(inverse ? 1.0 / s : s);
    }
}
__global__ __launch_bounds__(VEC_BLOCK) void k_f32_scale(float *__restrict__ x, int64_t n, double a)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) x[i] = (float)(a * (double)x[i]);
}
__global__ __launch_bounds__(VEC_BLOCK) void k_f32_axpy(float *__restrict__ y, const float *__restrict__ x, int64_t n, double a)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride)
        y[i] = (float)((double)y[i] + a * (double)x[i]);
}

}  // namespace lsqrhip

struct DevVecF {
    float *p = nullptr;
    ~DevVecF()
    {
        if (p) (void)hipFree(p);
    }
    int alloc(int64_t n)
    {
        hipError_t e = hipMalloc((void **)&p, sizeof(float) * (size_t)std::max<int64_t>(n, 1));
        return e == hipSuccess ? LSQRHIP_OK : fail(LSQRHIP_ERR_ALLOC, hipGetErrorString(e));
    }
};

// sum_i x_i y_i of two real32 device vectors, accumulated in binary64 in a fixed tree (x == y: the square of the
// norm -- real32 entries cannot overflow a binary64 square, so no scaling pass is needed)
static int f32_dot(H *h, int64_t n, const float *d_x, const float *d_y, double *result)
{
    *result = 0.0;
    if (n <= 0) return LSQRHIP_OK;
    const int g = vec_grid(n);
    hipLaunchKernelGGL(k_dot_T<float>, dim3(g), dim3(VEC_BLOCK), 0, h->stream, d_x, d_y, n, h->partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(VEC_BLOCK), 0, h->stream, (const double *)h->partials, g, h->d_scalar);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(result, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LSQRHIP_OK;
}
static int f32_nrm2(H *h, int64_t n, const float *d_x, double *result)
{
    RET(f32_dot(h, n, d_x, d_x, result));
    *result = std::sqrt(*result);
    return LSQRHIP_OK;
}
static int f32_scale(H *h, int64_t n, double a, float *d_x)
{
    if (n > 0) hipLaunchKernelGGL(k_f32_scale, dim3(vec_grid(n)), dim3(VEC_BLOCK), 0, h->stream, d_x, n, a);
    HIPCHK(hipGetLastError());
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_acheck_f32(lsqrhip_handle_t h, double eps, int *inform, double *relerr)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h || !inform) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!h->f32) return fail(LSQRHIP_ERR_ARG, "not a REAL32 handle (binary64 vectors on the device: lsqrhip_acheck)");
    HIPCHK(hipSetDevice(h->device));
    const int64_t m = h->m, n = h->n;
    hipStream_t s = h->stream;
    DevVecF v, w, x, y;
    RET(v.alloc(n)); RET(w.alloc(m)); RET(x.alloc(n)); RET(y.alloc(m));
    const double tol = std::pow(eps, 0.5);                                      // src/lsqr.f90:939
    hipLaunchKernelGGL(k_f32_acheck_fill, dim3(vec_grid(n)), dim3(VEC_BLOCK), 0, s, x.p, n, 0);   // :946-950
    hipLaunchKernelGGL(k_f32_acheck_fill, dim3(vec_grid(m)), dim3(VEC_BLOCK), 0, s, y.p, m, 1);   // :952-956
    double alfa = 0, beta = 0;
    RET(f32_nrm2(h, n, x.p, &alfa));                                            // :958-961
    RET(f32_nrm2(h, m, y.p, &beta));
    RET(f32_scale(h, n, 1.0 / alfa, x.p));
    RET(f32_scale(h, m, 1.0 / beta, y.p));
    if (m > 0) HIPCHK(hipMemcpyAsync(w.p, y.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToDevice, s));   // :969-972
    if (n > 0) HIPCHK(hipMemcpyAsync(v.p, x.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, s));
    RET(lsqrhip_aprod_device_f32(h, 1, x.p, w.p));
    RET(lsqrhip_aprod_device_f32(h, 2, v.p, y.p));
    RET(f32_dot(h, m, y.p, w.p, &alfa));                                        // :976-980
    RET(f32_dot(h, n, x.p, v.p, &beta));
    const double test1 = std::fabs(alfa - beta);
    const double test2 = 1.0 + std::fabs(alfa) + std::fabs(beta);
    const double test3 = test1 / test2;
    if (relerr) *relerr = test3;
    *inform = test3 <= tol ? 0 : 1;                                             // :984-992
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_xcheck_f32(lsqrhip_handle_t h, double anorm, double damp, double eps, const float *b,
                                  const float *x, float *u, float *v, float *w, int *inform, double *tests)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h || !inform || !tests) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!h->f32) return fail(LSQRHIP_ERR_ARG, "not a REAL32 handle (binary64 vectors on the device: lsqrhip_xcheck)");
    HIPCHK(hipSetDevice(h->device));
    const int64_t m = h->m, n = h->n;
    hipStream_t s = h->stream;
    DevVecF db, dx, du, dv, dw;
    RET(db.alloc(m)); RET(dx.alloc(n)); RET(du.alloc(m)); RET(dv.alloc(n)); RET(dw.alloc(n));
    if (m > 0) HIPCHK(hipMemcpyAsync(db.p, b, sizeof(float) * (size_t)m, hipMemcpyHostToDevice, s));
    if (n > 0) HIPCHK(hipMemcpyAsync(dx.p, x, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, s));
    const double dampsq = damp * damp, tol = std::pow(eps, 0.5);
    if (m > 0) HIPCHK(hipMemcpyAsync(du.p, db.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToDevice, s));   // :1073-1076
    RET(f32_scale(h, m, -1.0, du.p));
    RET(lsqrhip_aprod_device_f32(h, 1, dx.p, du.p));
    RET(f32_scale(h, m, -1.0, du.p));
    if (n > 0) HIPCHK(hipMemsetAsync(dv.p, 0, sizeof(float) * (size_t)n, s));  // :1080-1083
    RET(lsqrhip_aprod_device_f32(h, 2, dv.p, du.p));
    if (n > 0) HIPCHK(hipMemcpyAsync(dw.p, dv.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, s));   // :1089-1094
    if (damp != 0.0 && n > 0) {
        hipLaunchKernelGGL(k_f32_axpy, dim3(vec_grid(n)), dim3(VEC_BLOCK), 0, s, dw.p, (const float *)dx.p, n, -dampsq);
        HIPCHK(hipGetLastError());
    }
    double bnorm, xnorm, rho1, sigma1, rho2, sigma2;
    RET(f32_nrm2(h, m, db.p, &bnorm));                                          // :1098-1101
    RET(f32_nrm2(h, n, dx.p, &xnorm));
    RET(f32_nrm2(h, m, du.p, &rho1));
    RET(f32_nrm2(h, n, dv.p, &sigma1));
    if (damp == 0.0) {                                                          // :1110-1124
        rho2 = rho1;
        sigma2 = sigma1;
    } else {
        rho2 = std::sqrt(rho1 * rho1 + dampsq * (xnorm * xnorm));
        RET(f32_nrm2(h, n, dw.p, &sigma2));
    }
    double test1, test2, test3;
    if (bnorm == 0.0 && xnorm == 0.0) {                                         // :1129-1144
        *inform = 0;
        test1 = test2 = test3 = 0.0;
    } else {
        *inform = 4;
        test1 = rho1 / (bnorm + anorm * xnorm);
        test2 = 0.0;
        if (rho1 > 0.0) test2 = sigma1 / (anorm * rho1);
        test3 = test2;
        if (rho2 > 0.0) test3 = sigma2 / (anorm * rho2);
        if (test3 <= tol) *inform = 3;
        if (test2 <= tol) *inform = 2;
        if (test1 <= tol) *inform = 1;
    }
    tests[0] = test1;
    tests[1] = test2;
    tests[2] = test3;
    if (u && m > 0) HIPCHK(hipMemcpyAsync(u, du.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost, s));
    if (v && n > 0) HIPCHK(hipMemcpyAsync(v, dv.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s));
    if (w && n > 0) HIPCHK(hipMemcpyAsync(w, dw.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LSQRHIP_OK;
}
