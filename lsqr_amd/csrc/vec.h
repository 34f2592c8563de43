// vec.h -- the Golub-Kahan vector kernels (BLAS-1 shaped, HBM-bound).
//
//   k_update      x += t1 w ; w = t2 w + v ; dk += (t3 w)^2 ; [se += (t3 w)^2]
//                 replaces the inline loops at reference src/lsqr.f90:729-745, with the
//                 `dscal(n, 1/alpha, v)` of :697 folded in (v = V * sv).
//   k_sumsq       partial sums of x^2      (dnrm2, src/lsqrblas.f90:123-159; see note)
//   k_dot         partial sums of x*y      (ddot,  src/lsqrblas.f90:74-116)
//   k_scale       x <- a x                 (dscal, src/lsqrblas.f90:166-201)
//   k_copy        y <- x                   (dcopy, src/lsqrblas.f90:25-67)
//   k_copy_scale  w <- V * sv              (dscal + dcopy at src/lsqr.f90:642-643)
//   k_se_finish   se <- t * sqrt(se)       (src/lsqr.f90:857-865)
//
// Note on dnrm2: the reference's dlassq recurrence is a serial dependent chain
// with a divide per element; here the norm is sqrt(sum x^2) accumulated in a fixed
// tree.  Same value to a few ulp for |x| in ~[1e-150, 1e150] (LSQR's u, v are
// re-normalised every iteration, so they live at the scale of the matrix entries).
//
// All kernels: 256 threads, 16-byte (double2) accesses, capped grid-stride grid,
// one partial per workgroup reduced later in fixed order.
#pragma once

#include "common.h"
#include "csb.h"   // csb_hi_up
#include "state.h"

namespace lsqrhip {

constexpr int VEC_BLOCK = 256;
constexpr int VEC_MAX_GRID = 2048;

// Vectors are stored as VT = double, or float in the all-REAL32 build of the reference's precision macro
// (src/lsqr_kinds.F90:16-17): half the bytes of every vector pass.  Arithmetic is always binary64 in
// registers; a float vector holds the rounded results.
template <typename VT> struct Vec2;
template <> struct Vec2<double> { typedef double2 type; };
template <> struct Vec2<float> { typedef float2 type; };

// The work of ONE workgroup of the x/w update: block `ub` of a grid of `ugrid` blocks of
// VEC_BLOCK threads.  Returns (in thread 0) the block's partial of sum (t3 w)^2.  k_update
// runs it as a kernel of its own; the fused mode-1 kernels (spmv.h / sell.h, UpdArgs) run the
// same blocks from inside the SpMV launch -- same elements per thread, same order, same
// reduction, hence the same partials bit for bit.
template <typename VT, bool NT = false>
__device__ __forceinline__ double update_block(VT *__restrict__ x, VT *__restrict__ w,
                                               const VT *__restrict__ V, VT *__restrict__ se, int64_t n,
                                               double t1, double t2, double t3, double sv, bool wantse, int ub,
                                               int ugrid, double *red, VT *__restrict__ xc = nullptr)
{
    typedef typename Vec2<VT>::type V2T;
    double dk = 0.0;
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)ugrid * VEC_BLOCK;
    V2T *x2 = reinterpret_cast<V2T *>(x);
    V2T *w2 = reinterpret_cast<V2T *>(w);
    const V2T *V2 = reinterpret_cast<const V2T *>(V);
    V2T *se2 = reinterpret_cast<V2T *>(se);
    for (int64_t i = (int64_t)ub * VEC_BLOCK + threadIdx.x; i < n2; i += stride) {
        const V2T t = ld_stream2<NT>(&w2[i]);
        V2T xv = ld_stream2<NT>(&x2[i]);
        const V2T vv = V2[i];
        const double tx = (double)t.x, ty = (double)t.y;
        xv.x = (VT)(t1 * tx + (double)xv.x);
        xv.y = (VT)(t1 * ty + (double)xv.y);
        V2T wn;
        wn.x = (VT)(t2 * tx + (double)vv.x * sv);
        wn.y = (VT)(t2 * ty + (double)vv.y * sv);
        const double d0 = (t3 * tx) * (t3 * tx), d1 = (t3 * ty) * (t3 * ty);
        x2[i] = xv;
        w2[i] = wn;
        if (xc != nullptr) {   // (the tail of a batch: x also to where the solve was told to leave it -- element stores,
            xc[2 * i] = xv.x;  //  the caller's array need not be aligned for pairs)
            xc[2 * i + 1] = xv.y;
        }
        if (wantse) {
            V2T s = se2[i];
            s.x = (VT)(d0 + (double)s.x);
            s.y = (VT)(d1 + (double)s.y);
            se2[i] = s;
        }
        dk += d0;
        dk += d1;
    }
    if ((n & 1) && ub == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        const double t = (double)w[i];
        x[i] = (VT)(t1 * t + (double)x[i]);
        if (xc != nullptr) xc[i] = x[i];
        w[i] = (VT)(t2 * t + (double)V[i] * sv);
        const double d = (t3 * t) * (t3 * t);
        if (wantse) se[i] = (VT)(d + (double)se[i]);
        dk += d;
    }
    return block_sum<VEC_BLOCK>(dk, red);
}

template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_update(
    VT *__restrict__ x, VT *__restrict__ w, const VT *__restrict__ V,
    VT *__restrict__ se, int64_t n, const LsqrState *__restrict__ st,
    double *__restrict__ partials)
{
    if (st->stop != 0) return;
    const double t1 = st->t1, t2 = st->t2, t3 = st->t3, sv = st->sv;
    const bool wantse = st->wantse != 0;
    __shared__ double red[VEC_BLOCK / WAVE];
    const double tot = update_block(x, w, V, se, n, t1, t2, t3, sv, wantse, (int)blockIdx.x, (int)gridDim.x, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// w <- V * sv, the first w of the recurrence (src/lsqr.f90:641-644), in update_block's decomposition: what the
// FIRST lazy mode-1 launch of a solve carries in place of an x/w update (UpdArgs.on == 2).  The values are
// k_copy_scale's, bit for bit.
template <typename VT>
__device__ __forceinline__ void winit_block(VT *__restrict__ w, const VT *__restrict__ V, int64_t n, double sv, int ub,
                                            int ugrid)
{
    typedef typename Vec2<VT>::type V2T;
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)ugrid * VEC_BLOCK;
    V2T *w2 = reinterpret_cast<V2T *>(w);
    const V2T *V2 = reinterpret_cast<const V2T *>(V);
    for (int64_t i = (int64_t)ub * VEC_BLOCK + threadIdx.x; i < n2; i += stride) {
        const V2T vv = V2[i];
        V2T wn;
        wn.x = (VT)((double)vv.x * sv);
        wn.y = (VT)((double)vv.y * sv);
        w2[i] = wn;
    }
    if ((n & 1) && ub == 0 && threadIdx.x == 0) w[n - 1] = (VT)((double)V[n - 1] * sv);
}

// The x/w update of the PREVIOUS iteration carried by a lazy mode-1 SpMV launch ("fused
// update", solve_loop.h).  on = 0: nothing to do; on = 2: the first launch of a solve, w <- V sv instead.
struct UpdArgs {
    int on;
    int par;      // parity of the rotation inputs: st->rhobar2[par], st->phibar2[par]
    int ugrid;    // blocks of the update (== the grid k_update would use)
    int pad;
    void *x, *w, *se;   // VT arrays (double, or float for a REAL32 handle)
    const void *V;
    int64_t n;
    const LsqrState *st;
    const NormSlot *alpha_prev;  // (alpha, 1/alpha) of the previous iteration: used when beta == 0
    double *pout;                // update partials [ugrid]
};

// partials[b] = sum over this workgroup's share of x[i]*y[i]  (y == x: sum of squares)
__global__ __launch_bounds__(VEC_BLOCK) void k_dot(const double *__restrict__ x,
                                                   const double *__restrict__ y, int64_t n,
                                                   double *__restrict__ partials)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    double s = 0.0;
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    const double2 *y2 = reinterpret_cast<const double2 *>(y);
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n2; i += stride) {
        const double2 a = x2[i], b = y2[i];
        s += a.x * b.x;
        s += a.y * b.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) s += x[n - 1] * y[n - 1];
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// Blue's three range-safe accumulators of x^2 (scalar.h blue_add): partials[b], partials[g + b],
// partials[2 g + b] = this workgroup's small / mid / big sums, g = gridDim.x.  Same element order per
// thread as k_dot, so the mid plane of an in-range vector is k_dot(x, x)'s partials bit for bit.
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_sumsq3(const VT *__restrict__ x, int64_t n,
                                                      double *__restrict__ partials)
{
    typedef typename Vec2<VT>::type V2T;
    __shared__ double red[VEC_BLOCK / WAVE];
    Blue3 a{0.0, 0.0, 0.0};
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    const V2T *x2 = reinterpret_cast<const V2T *>(x);
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n2; i += stride) {
        const V2T v = x2[i];
        blue_add(a, (double)v.x);
        blue_add(a, (double)v.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) blue_add(a, (double)x[n - 1]);
    const double t0 = block_sum<VEC_BLOCK>(a.sml, red);
    const double t1 = block_sum<VEC_BLOCK>(a.med, red);
    const double t2 = block_sum<VEC_BLOCK>(a.big, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t0;
        partials[gridDim.x + blockIdx.x] = t1;
        partials[2 * gridDim.x + blockIdx.x] = t2;
    }
}

// The start of a solve in ONE pass (src/lsqr.f90:242, 621-637): u = b, v = x (= se) = 0 -- w is written whole by
// whoever forms the first w = v / alpha (k_copy_scale or winit_block), W may be null -- and the three Blue
// sums of b with k_sumsq3's decomposition over g = sgrid workgroups (same partials, bit for bit); the grid
// may be larger (zeroing n >> m elements).  b arrives through a slot in pinned host memory so that the launch
// can sit in a captured graph whatever address the caller passes; b == U (host b, already copied in): no store.
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_start(const void *const *__restrict__ bslot, VT *__restrict__ U, int64_t m,
                                                     int sgrid, VT *__restrict__ V, VT *__restrict__ X,
                                                     VT *__restrict__ W, VT *__restrict__ SE, int64_t n,
                                                     double *__restrict__ partials)
{
    typedef typename Vec2<VT>::type V2T;
    __shared__ double red[VEC_BLOCK / WAVE];
    __shared__ const void *bsh;
    // (the slot is read FIRST, by one lane of the workgroups that need it, and looked at LAST: a read across
    // PCIe, hidden behind the zeroing)
    const void *braw = nullptr;
    if (threadIdx.x == 0 && (int)blockIdx.x < sgrid) braw = *bslot;
    {   // zeros: every workgroup of the launch
        const int64_t n2 = n >> 1;
        const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
        V2T z;
        z.x = (VT)0;
        z.y = (VT)0;
        V2T *v2 = reinterpret_cast<V2T *>(V), *x2 = reinterpret_cast<V2T *>(X), *w2 = reinterpret_cast<V2T *>(W);
        V2T *s2 = reinterpret_cast<V2T *>(SE);
        for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n2; i += stride) {
            v2[i] = z;
            x2[i] = z;
            if (W != nullptr) w2[i] = z;
            if (SE != nullptr) s2[i] = z;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
            V[n - 1] = (VT)0;
            X[n - 1] = (VT)0;
            if (W != nullptr) W[n - 1] = (VT)0;
            if (SE != nullptr) SE[n - 1] = (VT)0;
        }
    }
    if ((int)blockIdx.x >= sgrid) return;
    if (threadIdx.x == 0) bsh = braw;
    __syncthreads();
    const VT *b = static_cast<const VT *>(bsh);
    const bool store = b != U;
    Blue3 a{0.0, 0.0, 0.0};
    const int64_t m2 = m >> 1;
    const int64_t stride = (int64_t)sgrid * VEC_BLOCK;
    const V2T *b2 = reinterpret_cast<const V2T *>(b);
    V2T *u2 = reinterpret_cast<V2T *>(U);
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < m2; i += stride) {
        const V2T v = b2[i];
        if (store) u2[i] = v;
        blue_add(a, (double)v.x);
        blue_add(a, (double)v.y);
    }
    if ((m & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const VT v = b[m - 1];
        if (store) U[m - 1] = v;
        blue_add(a, (double)v);
    }
    const double t0 = block_sum<VEC_BLOCK>(a.sml, red);
    const double t1 = block_sum<VEC_BLOCK>(a.med, red);
    const double t2 = block_sum<VEC_BLOCK>(a.big, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t0;
        partials[sgrid + blockIdx.x] = t1;
        partials[2 * sgrid + blockIdx.x] = t2;
    }
}

// out[0..2] = the three planes of k_sumsq3 reduced in fixed order (one workgroup): what the row-sharded
// solve all-reduces for norm(b)
__global__ __launch_bounds__(VEC_BLOCK) void k_reduce_partials3(const double *__restrict__ partials, int np,
                                                                double *__restrict__ out)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    for (int k = 0; k < 3; ++k) {
        const double s = np > 0 ? strided_sum<VEC_BLOCK>(partials + (size_t)k * np, np) : 0.0;
        const double tot = block_sum<VEC_BLOCK>(s, red);
        if (threadIdx.x == 0) out[k] = tot;
    }
}

__global__ __launch_bounds__(VEC_BLOCK) void k_scale(double *__restrict__ x, int64_t n, double a)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) x[i] = a * x[i];
}

__global__ __launch_bounds__(VEC_BLOCK) void k_copy(const double *__restrict__ x,
                                                    double *__restrict__ y, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) y[i] = x[i];
}

// y <- x, `bytes` bytes: 16 at a time when both addresses allow it, else 4 (vectors of float / double)
__global__ __launch_bounds__(VEC_BLOCK) void k_copy_bytes(const void *__restrict__ x, void *__restrict__ y, int64_t bytes)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    const int64_t t = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        const int64_t n16 = bytes >> 4;
        const uint4 *a = static_cast<const uint4 *>(x);
        uint4 *b = static_cast<uint4 *>(y);
        for (int64_t i = t; i < n16; i += stride) b[i] = a[i];
        const unsigned *a4 = static_cast<const unsigned *>(x);
        unsigned *b4 = static_cast<unsigned *>(y);
        for (int64_t i = (n16 << 2) + t; i < (bytes >> 2); i += stride) b4[i] = a4[i];
    } else {
        const unsigned *a4 = static_cast<const unsigned *>(x);
        unsigned *b4 = static_cast<unsigned *>(y);
        for (int64_t i = t; i < (bytes >> 2); i += stride) b4[i] = a4[i];
    }
}

// The copy out of x at the end of a device-resident solve, as the last node of every graph batch: does nothing
// until the stop flag is up (k_s3 raises it at itnlim at the latest), then x is final and goes to the address
// the solve was given (LsqrState.xout; null: the host copies).  No host round trip between the last
// iteration and the copy.
__global__ __launch_bounds__(VEC_BLOCK) void k_out_copy(const LsqrState *__restrict__ st, const void *__restrict__ X,
                                                        int64_t bytes)
{
    if (st->stop == 0 || st->xout == nullptr) return;
    void *y = st->xout;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    const int64_t t = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x;
    if ((((uintptr_t)X | (uintptr_t)y) & 15) == 0) {
        const int64_t n16 = bytes >> 4;
        const uint4 *a = static_cast<const uint4 *>(X);
        uint4 *b = static_cast<uint4 *>(y);
        for (int64_t i = t; i < n16; i += stride) b[i] = a[i];
        const unsigned *a4 = static_cast<const unsigned *>(X);
        unsigned *b4 = static_cast<unsigned *>(y);
        for (int64_t i = (n16 << 2) + t; i < (bytes >> 2); i += stride) b4[i] = a4[i];
    } else {
        const unsigned *a4 = static_cast<const unsigned *>(X);
        unsigned *b4 = static_cast<unsigned *>(y);
        for (int64_t i = t; i < (bytes >> 2); i += stride) b4[i] = a4[i];
    }
}

__global__ __launch_bounds__(VEC_BLOCK) void k_fill(double *__restrict__ x, int64_t n, double a)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) x[i] = a;
}

// w <- V * sv  (first w of the recurrence, src/lsqr.f90:641-644)
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_copy_scale(VT *__restrict__ w,
                                                          const VT *__restrict__ V, int64_t n,
                                                          const LsqrState *__restrict__ st)
{
    if (st->stop != 0) return;
    const double sv = st->sv;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) w[i] = (VT)((double)V[i] * sv);
}

// y <- y + a*x   (xcheck's w = w - damp^2 x, src/lsqr.f90:1090-1094)
__global__ __launch_bounds__(VEC_BLOCK) void k_axpy(double *__restrict__ y,
                                                    const double *__restrict__ x, int64_t n, double a)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride)
        y[i] = y[i] + a * x[i];
}

// acheck's "unlikely" vectors (src/lsqr.f90:946-956): x_j = sqrt(j+1), y_i = 1/sqrt(i+1), 1-based j,i
__global__ __launch_bounds__(VEC_BLOCK) void k_acheck_fill(double *__restrict__ x, int64_t n, int inverse)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) {
        const double s = sqrt((double)(i + 2));
        x[i] = inverse ? 1.0 / s : s;
    }
}

// se <- t * sqrt(se), t = rnorm / sqrt(1 | m-n | m)   (src/lsqr.f90:857-865)
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_se_finish(VT *__restrict__ se, int64_t n,
                                                         const LsqrState *__restrict__ st)
{
    if (st->wantse == 0 || st->itn == 0) return;
    double t = 1.0;
    if (st->m > st->n) t = (double)(st->m - st->n);
    if (st->damped) t = (double)st->m;
    t = st->rnorm / sqrt(t);
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride)
        se[i] = (VT)(t * sqrt((double)se[i]));
}

// partials[b] = max |x[i]| over this workgroup's share (first pass of the scaled norm)
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_amax(const VT *__restrict__ x, int64_t n,
                                                    double *__restrict__ partials)
{
    __shared__ double red[VEC_BLOCK];
    double m = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) m = fmax(m, fabs((double)x[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = VEC_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// The max|x| pass in front of a column-swept product (csb.h), under a name of its own so that profiles and PMC
// passes can tell it from the build's k_amax over the matrix values.  One workgroup per PIECE of x (csb.h csb_pieces:
// aligned groups of 2^L elements dealt round-robin to NP pieces), one maximum each: from them csb.h takes max|x| and --
// the median piece -- what the bulk of x looks like (its "tau").
// A piece's maximum is kept to the high word of the binary64, rounded UP (csb_hi_up): a bound within 2^-19 of the
// maximum, which a product that WRITES the vector can also keep with a 32-bit wave reduction per 64 rows and one atomic
// max per group (csb.h csb_group_max) -- inside the solver's loop the pass is left out and both ways give the same words.
constexpr int CSB_XMAX_GRID = 1024;   // (x 4: the most pieces, csb.h CSB_XMAX_PIECES)
static_assert(CSB_XMAX_GRID * (VEC_BLOCK / WAVE) == CSB_XMAX_PIECES, "xmax_part holds one word per piece");
static inline int csb_npieces(int64_t n) { return csb_pieces(n).NP; }
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_csb_xmax(const VT *__restrict__ x, int64_t n, CsbPieces pc,
                                                        double *__restrict__ partials)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    double m = 0.0;
    const int64_t groups = (n + ((int64_t)1 << pc.L) - 1) >> pc.L;
    for (int64_t g = blockIdx.x; g < groups; g += pc.NP) {
        const int64_t i0 = g << pc.L, i1 = std::min<int64_t>(i0 + ((int64_t)1 << pc.L), n);
        for (int64_t i = i0 + threadIdx.x; i < i1; i += VEC_BLOCK) m = fmax(m, fabs((double)x[i]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, WAVE));
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < VEC_BLOCK / WAVE; ++i) m = fmax(m, red[i]);
        partials[blockIdx.x] = __longlong_as_double((long long)((unsigned long long)csb_hi_up(m) << 32));
    }
}

// partials[b] = sum over this workgroup's share of (x[i] * sc)^2   (sc a power of two: exact)
__global__ __launch_bounds__(VEC_BLOCK) void k_sumsq_scaled(const double *__restrict__ x, int64_t n, double sc,
                                                            double *__restrict__ partials)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) {
        const double t = x[i] * sc;
        s += t * t;
    }
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// y <- (T) x elementwise (REAL32 handles: values and vectors converted at the boundary / after the build)
template <typename TI, typename TO>
__global__ __launch_bounds__(VEC_BLOCK) void k_convert(const TI *__restrict__ x, TO *__restrict__ y, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) y[i] = (TO)x[i];
}

// out[0] = sum of partials[0..np) in fixed order (one workgroup).
__global__ __launch_bounds__(VEC_BLOCK) void k_reduce_partials(const double *__restrict__ partials,
                                                               int np, double *__restrict__ out)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    const double s = np > 0 ? strided_sum<VEC_BLOCK>(partials, np) : 0.0;
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) out[0] = tot;
}

}  // namespace lsqrhip
