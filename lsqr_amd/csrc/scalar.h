// scalar.h -- the scalar recurrences of LSQR as one-workgroup kernels.
//
// Replaces reference src/lsqr.f90:597-617, 632-653 (initialisation), :687-689
// (anorm), :703-721 (damping + QR rotation), :751-810 (estimates, stopping tests),
// :843-850 (nconv rule), :871 (istop 2 -> 3) and d2norm (:1164-1179).
//
// Each kernel first reduces the per-workgroup partials of a vector kernel in a fixed
// order (template REDUCE = true), or takes an already reduced / all-reduced sum
// (REDUCE = false, multi-GPU), then thread 0 runs the scalar work and publishes the
// coefficients the next vector kernel needs.  Nothing leaves the device during the loop.
//
// In the pipelined single-GPU schedule (solve_loop.h) these kernels are OFF the critical
// path: k_s12 (steps 1+2 merged) and k_s3 run on a side stream next to the following SpMV,
// which derives the one norm it needs by itself (spmv.h, lazy coefficients).
#pragma once

#include "common.h"
#include "state.h"

namespace lsqrhip {

constexpr int SC_BLOCK = 256;

__device__ __forceinline__ double d2norm(double a, double b)  // src/lsqr.f90:1164-1179
{
    const double scale = fabs(a) + fabs(b);
    if (scale == 0.0) return 0.0;
    const double p = a / scale, q = b / scale;
    return scale * sqrt(p * p + q * q);
}

template <bool REDUCE>
__device__ __forceinline__ double take_sum(const double *partials, int np, const double *pre, double *red)
{
    if (!REDUCE) return *pre;
    const double s = np > 0 ? strided_sum<SC_BLOCK>(partials, np) : 0.0;
    return block_sum<SC_BLOCK>(s, red);  // valid in thread 0
}

// ---- range-safe norms (dnrm2, src/lsqrblas.f90:123-159) ----------------------------------------
// The reference's dnrm2 is the scaled (dlassq) recurrence: it cannot over- or underflow.  Two
// forms here, both one pass over the data and both with the reference's full range:
//
//  * fused in-loop norms of a MATRIX handle: the producing kernel sums (y * ns)^2 with ns a power
//    of two fixed per matrix (2^-e, 2^e ~ max|a_ij|: u = A v - alpha u and v = A'u - beta v live at
//    the scale of the matrix entries whatever b is), and every consumer takes sqrt(sum) * (1/ns).
//    Scaling by a power of two is exact, so within the old range nothing changes by a bit.
//  * norm(b) and every norm of a user OPERATOR (no matrix to take a scale from): Blue's three
//    accumulators as in LAPACK 3.10's dnrm2 -- squares of big / small elements are summed in
//    scaled form, the rest plainly -- combined by blue_norm.  Within the mid range (1.5e-154 <
//    |x| < 2e146 for all elements) this is the plain sqrt(sum x^2), bit for bit.
// The norm a scalar step starts from (valid in thread 0).  BLUE = false: sqrt(sum of the np partials
// of (y * ns)^2) * ninv, or of the already reduced *pre.  BLUE = true: three planes of np partials
// (small, mid, big) at partials, partials + np, partials + 2 np -- or pre[0..2].
template <bool REDUCE, bool BLUE>
__device__ __forceinline__ double take_norm(const double *partials, int np, const double *pre, double ninv,
                                            double *red)
{
    if (!BLUE) return sqrt(take_sum<REDUCE>(partials, np, pre, red)) * ninv;
    const double a0 = take_sum<REDUCE>(partials, np, pre, red);
    const double a1 = take_sum<REDUCE>(partials + np, np, pre + 1, red);
    const double a2 = take_sum<REDUCE>(partials + 2 * (size_t)np, np, pre + 2, red);
    return blue_norm(a0, a1, a2);
}

// ---- step 1: after mode 1.  beta = norm(A v - alpha u); anorm        (:675, :683-693) ----
__device__ __forceinline__ void s1_step(LsqrState *st, double beta)
{
    st->itn = st->itn + 1;
    const double alpha = st->alpha;
    st->beta = beta;
    double temp = d2norm(alpha, beta);
    temp = d2norm(temp, st->damp);
    st->anorm = d2norm(st->anorm, temp);
    if (beta > 0.0) {
        st->su = 1.0 / beta;
        st->c2.sx = st->su;
        st->c2.sy = st->sv;
        st->c2.cy = -beta;
        st->c2.skip = 0;
    } else {
        st->su = 1.0;
        st->c2.skip = 1;
    }
    st->c2p.sx = st->su;  // sharded: T <- 0*(T*1) + A_p'(U su); V <- c2.cy (V c2.sy) + sum_p T_p
    st->c2p.sy = 1.0;
    st->c2p.cy = 0.0;
    st->c2p.skip = st->c2.skip;
}

// The damping rotation and the QR rotation of one iteration (src/lsqr.f90:703-721) and the
// coefficients of the x/w update (:729-745).  A pure function of its arguments: the scalar
// machine (s2_step) and every workgroup of a fused mode-1 + update kernel evaluate it on the
// same inputs and get the same bits.
struct Rot {
    double psi, phibar, rhobar, rho, phi, theta, tau, t1, t2, t3;
};
__device__ __forceinline__ Rot rot_step(double rhobar, double phibar, double damp, int damped, double alpha,
                                        double beta)
{
    Rot r;
    double rhbar1 = rhobar;
    r.psi = 0.0;
    if (damped) {
        rhbar1 = d2norm(rhobar, damp);
        const double cs1 = rhobar / rhbar1;
        const double sn1 = damp / rhbar1;
        r.psi = sn1 * phibar;
        phibar = cs1 * phibar;
    }
    const double rho = d2norm(rhbar1, beta);
    const double cs = rhbar1 / rho;
    const double sn = beta / rho;
    const double theta = sn * alpha;
    r.rhobar = -cs * alpha;
    const double phi = cs * phibar;
    r.phibar = sn * phibar;
    r.tau = sn * phi;
    r.rho = rho;
    r.phi = phi;
    r.theta = theta;
    r.t1 = phi / rho;
    r.t2 = -theta / rho;
    r.t3 = 1.0 / rho;
    return r;
}

// ---- step 2: after mode 2.  alpha = norm(A'u - beta v); rotations; update coefficients (:695-726)
__device__ __forceinline__ void s2_step(LsqrState *st, double alpha_new, bool skipped)
{
    double alpha = st->alpha;
    const double beta = st->beta;
    if (!skipped) {
        alpha = alpha_new;
        st->alpha = alpha;
        st->sv = alpha > 0.0 ? 1.0 / alpha : 1.0;
    }
    const int k = st->itn & 1;  // s1_step has already advanced itn to this iteration
    const Rot r = rot_step(st->rhobar2[k ^ 1], st->phibar2[k ^ 1], st->damp, st->damped, alpha, beta);
    if (st->damped) st->psi = r.psi;
    st->rhobar2[k] = r.rhobar;
    st->phibar2[k] = r.phibar;
    st->tau = r.tau;
    st->rho = r.rho;
    st->phi = r.phi;
    st->theta = r.theta;
    st->t1 = r.t1;
    st->t2 = r.t2;
    st->t3 = r.t3;
    st->c1.sx = st->sv;
    st->c1.sy = st->su;
    st->c1.cy = -alpha;
    st->c1.skip = 0;
}

// after the three Blue sums of b (vec.h k_sumsq3): beta = norm(b); u = b/beta     (src/lsqr.f90:597-617, 632-637)
// `slot` (optional) receives (beta, 1/beta) for the first pipelined mode-1 launch.
// `init` (optional): the solve's initial state in pinned host memory, copied into *st first (saves a copy node).
template <bool REDUCE>
__global__ __launch_bounds__(SC_BLOCK) void k_s_init1(const double *partials, int np, const double *pre,
                                                      LsqrState *st, NormSlot *slot, const LsqrState *init = nullptr)
{
    __shared__ double red[SC_BLOCK / WAVE];
    if (init != nullptr) {
        static_assert(sizeof(LsqrState) % sizeof(int) == 0, "copied by words");
        const int *src = reinterpret_cast<const int *>(init);
        int *dst = reinterpret_cast<int *>(st);
        for (int i = threadIdx.x; i < (int)(sizeof(LsqrState) / sizeof(int)); i += SC_BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    const double beta = take_norm<REDUCE, true>(partials, np, pre, 1.0, red);  // norm(b): Blue's form
    if (threadIdx.x != 0) return;
    st->beta = beta;
    st->alpha = 0.0;
    st->sv = 1.0;
    if (beta > 0.0) {
        st->su = 1.0 / beta;
        st->c2.sx = st->su;  // V <- 0*(V*1) + A'(U*su), V pre-zeroed
        st->c2.sy = 1.0;
        st->c2.cy = 0.0;
        st->c2.skip = 0;
    } else {
        st->su = 1.0;
        st->c2.skip = 1;
    }
    st->c2p = st->c2;  // sharded: T <- 0*(T*1) + A_p'(U su)
    if (slot) {
        slot->nrm = beta;
        slot->scale = st->su;
    }
}

// after sum(V^2): alpha = norm(A'u); v = V/alpha; arnorm; loop entry     (:638-653)
__device__ __forceinline__ void s_init2_step(LsqrState *st, double nrm)
{
    const bool skipped = st->c2.skip != 0;
    const double beta = st->beta;
    const double alpha = skipped ? 0.0 : nrm;
    st->alpha = alpha;
    st->sv = alpha > 0.0 ? 1.0 / alpha : 1.0;
    st->arnorm = alpha * beta;
    // the reference leaves these unassigned when the loop is skipped; define them
    st->bnorm = beta;
    st->rnorm = beta;
    st->alpha0 = alpha;
    st->beta0 = beta;
    st->test2_0 = beta > 0.0 ? alpha / beta : 0.0;
    if (st->arnorm == 0.0) {
        st->stop = 1;  // istop stays 0: x = 0 is the exact solution
        return;
    }
    st->rhobar2[0] = alpha;  // itn = 0
    st->phibar2[0] = beta;
    st->c1.sx = st->sv;
    st->c1.sy = st->su;
    st->c1.cy = -alpha;
    st->c1.skip = 0;
}

template <bool REDUCE, bool BLUE = false>
__global__ __launch_bounds__(SC_BLOCK) void k_s_init2(const double *partials, int np, const double *pre,
                                                      LsqrState *st)
{
    __shared__ double red[SC_BLOCK / WAVE];
    const double nrm = take_norm<REDUCE, BLUE>(partials, np, pre, st->ns_inv, red);
    if (threadIdx.x != 0) return;
    s_init2_step(st, nrm);
}

template <bool REDUCE, bool BLUE = false>
__global__ __launch_bounds__(SC_BLOCK) void k_s1(const double *partials, int np, const double *pre, LsqrState *st)
{
    if (st->stop != 0) return;
    __shared__ double red[SC_BLOCK / WAVE];
    const double nrm = take_norm<REDUCE, BLUE>(partials, np, pre, st->ns_inv, red);
    if (threadIdx.x != 0) return;
    s1_step(st, nrm);
}

template <bool REDUCE, bool BLUE = false>
__global__ __launch_bounds__(SC_BLOCK) void k_s2(const double *partials, int np, const double *pre, LsqrState *st)
{
    if (st->stop != 0) return;
    __shared__ double red[SC_BLOCK / WAVE];
    const bool skipped = st->c2.skip != 0;
    const double nrm = take_norm<REDUCE, BLUE>(partials, np, pre, st->ns_inv, red);
    if (threadIdx.x != 0) return;
    s2_step(st, nrm, skipped);
}

// steps 1 and 2 in one launch: p1 = partials of the mode-1 SpMV, p2 = partials of the
// mode-2 SpMV of the same iteration (pipelined schedule; both products have already run).
__global__ __launch_bounds__(SC_BLOCK) void k_s12(const double *p1, int np1, const double *p2, int np2,
                                                  LsqrState *st)
{
    if (st->stop != 0) return;
    __shared__ double red[SC_BLOCK / WAVE];
    const double sum1 = take_sum<true>(p1, np1, nullptr, red);
    const double sum2 = take_sum<true>(p2, np2, nullptr, red);
    if (threadIdx.x != 0) return;
    s1_step(st, sqrt(sum1) * st->ns_inv);
    s2_step(st, sqrt(sum2) * st->ns_inv, st->c2.skip != 0);
}

// ---- step 3: after the x/w update.  dknorm, norm estimates, stopping tests, istop (:751-810, 843-850)
// x(1) for the log, from a vector stored as double or (REAL32 handle) float
__device__ __forceinline__ double first_of(const void *x, int f32)
{
    if (x == nullptr) return 0.0;
    return f32 ? (double)static_cast<const float *>(x)[0] : static_cast<const double *>(x)[0];
}

__device__ __forceinline__ void s3_step(LsqrState *st, double sum, double x1, double *log)
{
    const int itn = st->itn;
    const double rho = st->rho, phi = st->phi, theta = st->theta, tau = st->tau;
    const double alpha = st->alpha;

    const double dknorm = sqrt(sum);
    const double dnorm = d2norm(st->dnorm, dknorm);
    st->dnorm = dnorm;
    const double dxk = fabs(phi * dknorm);
    if (st->dxmax < dxk) {
        st->dxmax = dxk;
        st->maxdx = itn;
    }

    const double delta = st->sn2 * rho;
    const double gambar = -st->cs2 * rho;
    const double rhs = phi - delta * st->z;
    const double zbar = rhs / gambar;
    const double xnorm = d2norm(st->xnorm1, zbar);
    const double gamma = d2norm(gambar, theta);
    st->cs2 = gambar / gamma;
    st->sn2 = theta / gamma;
    st->z = rhs / gamma;
    st->xnorm1 = d2norm(st->xnorm1, st->z);
    st->xnorm = xnorm;

    const double anorm = st->anorm, bnorm = st->bnorm;
    const double acond = anorm * dnorm;
    st->acond = acond;
    st->res2 = d2norm(st->res2, st->psi);
    const double rnorm = d2norm(st->res2, st->phibar2[itn & 1]);
    st->rnorm = rnorm;
    const double arnorm = alpha * fabs(tau);
    st->arnorm = arnorm;

    const double alfopt = sqrt(rnorm / (dnorm * xnorm));
    const double test1 = rnorm / bnorm;
    double test2 = 0.0;
    if (rnorm > 0.0) test2 = arnorm / (anorm * rnorm);
    const double test3 = 1.0 / acond;
    double t1 = test1 / (1.0 + anorm * xnorm / bnorm);
    const double rtol = st->btol + st->atol * anorm * xnorm / bnorm;

    double t3 = 1.0 + test3;
    double t2 = 1.0 + test2;
    t1 = 1.0 + t1;
    if (st->wp32) {
        // wp = real32 (src/lsqr_kinds.F90:16-17): the reference forms these three sums in real32, i.e. the
        // "machine precision" stops (src/lsqr.f90:795-797) fire at eps(real32) -- all that real32 vectors can reach
        t3 = (double)(float)t3;
        t2 = (double)(float)t2;
        t1 = (double)(float)t1;
    }
    int istop = st->istop;
    if (itn >= st->itnlim) istop = 5;
    if (t3 <= 1.0) istop = 4;
    if (t2 <= 1.0) istop = 2;
    if (t1 <= 1.0) istop = 1;
    if (test3 <= st->ctol) istop = 4;
    if (test2 <= st->atol) istop = 2;
    if (test1 <= rtol) istop = 1;

    // Keep a record only for the iterations the reference prints (src/lsqr.f90:815-822), so the
    // buffer stays small even for itnlim = 4(m+n+50).
    const bool show = (st->n <= 40) || (itn <= 10) || (itn >= st->itnlim - 10) || (itn % 10 == 0) ||
                      (test3 <= 2.0 * st->ctol) || (test2 <= 10.0 * st->atol) || (test1 <= 10.0 * rtol) ||
                      (istop != 0);
    // A run that sits inside the "near convergence" bands for long can print more lines than the buffer
    // holds: the last slot is kept for the record of the stopping iteration (the line every reader looks
    // for), whatever had to be dropped is flagged (option "log_truncated").
    int slot = -1;
    if (st->want_log && show) {
        if (st->log_count < st->log_cap - 1 || (istop != 0 && st->log_count < st->log_cap)) {
            slot = st->log_count;
            st->log_count = st->log_count + 1;
        } else {
            st->log_truncated = 1;
            if (istop != 0 && st->log_cap > 0) slot = st->log_cap - 1;
        }
    }
    if (slot >= 0) {
        double *r = log + (size_t)slot * LOG_STRIDE;
        r[0] = (double)itn; r[1] = x1; r[2] = rnorm; r[3] = test1; r[4] = test2; r[5] = anorm;
        r[6] = acond; r[7] = phi; r[8] = dknorm; r[9] = dxk; r[10] = alfopt; r[11] = (double)istop;
        r[12] = rtol; r[13] = xnorm;
    }

    if (istop == 0) {
        st->nstop = 0;
    } else {
        const int nconv = 1;
        st->nstop = st->nstop + 1;
        if (st->nstop < nconv && itn < st->itnlim) istop = 0;
    }
    st->istop = istop;
    if (istop != 0) st->stop = 1;
}

template <bool REDUCE>
__global__ __launch_bounds__(SC_BLOCK) void k_s3(const double *partials, int np, const double *pre,
                                                 LsqrState *st, const void *x, int xf32, double *log)
{
    if (st->stop != 0) return;
    __shared__ double red[SC_BLOCK / WAVE];
    const double sum = take_sum<REDUCE>(partials, np, pre, red);
    if (threadIdx.x != 0) return;
    s3_step(st, sum, first_of(x, xf32), log);
}

// k_s3 at the end of a batch, followed by the snapshot the host polls: the state as it now is -- settled by this
// very kernel, or long stopped -- written straight into one of two slots in pinned host memory (alternating by
// batch: the look-ahead poll has two batches in flight).  Replaces a device-to-host copy command behind the
// batch (~13 us of gap + copy at the end of every batch of a short solve).
template <bool REDUCE>
__global__ __launch_bounds__(SC_BLOCK) void k_s3_snap(const double *partials, int np, const double *pre,
                                                      LsqrState *st, const void *x, int xf32, double *log,
                                                      LsqrState *snap)
{
    __shared__ double red[SC_BLOCK / WAVE];
    if (st->stop == 0) {  // uniform
        const double sum = take_sum<REDUCE>(partials, np, pre, red);
        if (threadIdx.x == 0) s3_step(st, sum, first_of(x, xf32), log);
    }
    __syncthreads();
    const int k = st->batch;
    const int *src = reinterpret_cast<const int *>(st);
    int *dst = reinterpret_cast<int *>(snap + (k & 1));
    for (int i = threadIdx.x; i < (int)(sizeof(LsqrState) / sizeof(int)); i += SC_BLOCK) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        st->batch = k + 1;
        // the seal, last: every word of the snapshot is performed system-wide (fence + barrier above) before this store
        // is issued, so a host that SPINS on it (solve_loop.h: ~10 us sooner than hipEventSynchronize wakes up) and
        // then reads the slot reads the settled state
        __hip_atomic_store(&(snap + (k & 1))->seal, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- riders -----------------------------------------------------------------------------
// The same scalar steps executed by ONE spare workgroup of a long vector kernel (spmv.h):
// the rider's inputs are complete when its host kernel starts and its outputs are first read
// by a LATER kernel, so kernel-boundary ordering is all the synchronisation there is.
struct Rider {
    int kind;            // 0 none, 1 = steps 1+2 (pa = mode-1 partials, pb = mode-2 partials), 2 = step 3 (pa),
                         // 3 = k_s_init2 (pa = the mode-2 partials of the start of the solve)
    int na, nb;
    const double *pa, *pb;
    LsqrState *st;
    const void *x;       // step 3: x(1) for the log
    int xf32;            //         ... stored as float (REAL32 handle)
    double *log;
};

__device__ __forceinline__ void run_rider(const Rider &r, double *red)
{
    LsqrState *st = r.st;
    if (r.kind == 3) {   // (runs whatever the stop flag says, like the kernel it stands for)
        const double sum = take_sum<true>(r.pa, r.na, nullptr, red);
        if (threadIdx.x != 0) return;
        s_init2_step(st, sqrt(sum) * st->ns_inv);
        return;
    }
    if (st->stop != 0) return;
    if (r.kind == 1) {
        const double sum1 = take_sum<true>(r.pa, r.na, nullptr, red);
        const double sum2 = take_sum<true>(r.pb, r.nb, nullptr, red);
        if (threadIdx.x != 0) return;
        s1_step(st, sqrt(sum1) * st->ns_inv);
        s2_step(st, sqrt(sum2) * st->ns_inv, st->c2.skip != 0);
    } else if (r.kind == 2) {
        const double sum = take_sum<true>(r.pa, r.na, nullptr, red);
        if (threadIdx.x != 0) return;
        s3_step(st, sum, first_of(r.x, r.xf32), r.log);
    }
}

}  // namespace lsqrhip
