// common.h -- shared device helpers for the LSQR HIP kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsqrhip {

constexpr int WAVE = 64;

// store_through: a store that goes on to memory (system scope: sc0 sc1) while the kernel runs instead of waiting in
// L2 for the write-back at the kernel's end.  Used for y of the row-pattern product (pat.h): a launch of ~7 us that
// leaves 8 MB dirty in the L2s otherwise spends ~0.7 us of its life on that write-back with nothing else to do --
// 7.57 -> 6.88 us per product, 44.1k -> 46.8k iterations/s at configs[1]; no difference at HBM-resident sizes.
// Measured and NOT used (profiles/r03/config2_patterns.txt): for x and w of the update (the update alone 5.3 -> 6.5 us,
// the gain above gone), and for y of the packed-record product of sell.h (alone 7.7 -> 7.2 us, the solve 40.6k -> 39.1k).
typedef double lsqrhip_d2 __attribute__((ext_vector_type(2)));
typedef float lsqrhip_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_through(double *p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_through(float *p, float v)
{
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// ld_stream<NT>: a load of matrix data that is read once per product.  Non-temporal (NT) when an iteration's working
// set exceeds twice the 256 MB Infinity Cache (Csr.nt, decided at create): the stream then does not displace what is
// reused (x, y) from the caches -- 16M-row 5-point operator: packed records 124 -> 98 us per product, structure
// patterns 214 -> 172 us.  Plain when it fits: there the matrix itself is what the caches hold from one product to the
// next (1M rows, non-temporal: 31.2k -> 28.0k iterations/s).  A compile-time choice: as a run-time flag the two
// forms of every load cost the 1M-row kernels 2 us per launch.  profiles/r03/config2_patterns.txt section 14.
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}
typedef double lsqrhip_d2v __attribute__((ext_vector_type(2)));
typedef float lsqrhip_f2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld_stream2(const double2 *p)
{
    if (!NT) return *p;
    const lsqrhip_d2v q = __builtin_nontemporal_load(reinterpret_cast<const lsqrhip_d2v *>(p));
    return make_double2(q.x, q.y);
}
template <bool NT>
__device__ __forceinline__ float2 ld_stream2(const float2 *p)
{
    if (!NT) return *p;
    const lsqrhip_f2v q = __builtin_nontemporal_load(reinterpret_cast<const lsqrhip_f2v *>(p));
    return make_float2(q.x, q.y);
}
typedef unsigned lsqrhip_u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld_stream4(const uint4 *p)
{
    if (!NT) return *p;
    const lsqrhip_u4 q = __builtin_nontemporal_load(reinterpret_cast<const lsqrhip_u4 *>(p));
    return make_uint4(q.x, q.y, q.z, q.w);
}

// Fixed-shape reductions: the shuffle tree and the cross-wave order are the same
// on every launch, so every norm is reproducible run to run (istop / itn depend
// on them, reference src/lsqr.f90:635, 641, 691, 696, 798-810).
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;  // lane 0 holds the sum
}

// Sum over the workgroup; result valid in thread 0.  `lds` holds BLOCK/64 doubles.
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v, double *lds)
{
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < BLOCK / WAVE; ++i) r += lds[i];
    }
    return r;
}

// This thread's share of a fixed-order sum of np partials: p[tid] + p[tid+BLOCK] + ... in that
// order.  The first 8 loads are issued together (np <= 2048 = every grid cap in this library),
// because a plain `for (...) s += p[i]` waits for each load in turn: 8 dependent L2 round
// trips, ~4 us of every scalar kernel and every lazy SpMV prologue before this was unrolled.
template <int BLOCK, int K>
__device__ __forceinline__ double strided_sum_k(const double *__restrict__ p, int np)
{
    const int t = threadIdx.x;
    const int last = np > 0 ? np - 1 : 0;
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = t + k * BLOCK;
        v[k] = p[i < last ? i : last];
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (t + k * BLOCK < np) s += v[k];
    for (int i = t + K * BLOCK; i < np; i += BLOCK) s += p[i];
    return s;
}
// Only as many loads as np needs (uniform branch): every workgroup of a lazy product runs this,
// and at 2048 workgroups the loads themselves are the cost (scripts/sell_roof.hip "prologue").
template <int BLOCK>
__device__ __forceinline__ double strided_sum(const double *__restrict__ p, int np)
{
    if (np <= 2 * BLOCK) return strided_sum_k<BLOCK, 2>(p, np);
    if (np <= 4 * BLOCK) return strided_sum_k<BLOCK, 4>(p, np);
    return strided_sum_k<BLOCK, 8>(p, np);
}

// Fixed-order sum of np doubles by the whole workgroup, result returned to EVERY thread.
// Same per-thread stride, shuffle tree and cross-wave order as the scalar kernels use
// (scalar.h take_sum), so any kernel reducing the same partials obtains the same bits.
// `lds` holds BLOCK/64 + 1 doubles.
template <int BLOCK>
__device__ __forceinline__ double block_sum_all(const double *__restrict__ p, int np, double *lds)
{
    const double s = np > 0 ? strided_sum<BLOCK>(p, np) : 0.0;
    const double r = block_sum<BLOCK>(s, lds);
    if (threadIdx.x == 0) lds[BLOCK / WAVE] = r;
    __syncthreads();
    const double out = lds[BLOCK / WAVE];
    __syncthreads();  // lds may be reused right away by the caller
    return out;
}

// block_sum_all for a caller that requested its share of the partials itself (strided_share_load): the loads were
// issued early, among other requests, and are summed here in strided_sum's order -- the same bits.
template <int BLOCK, int SHARE_K>
__device__ __forceinline__ void strided_share_load(const double *__restrict__ p, int np, double (&v)[SHARE_K])
{
    const int t = threadIdx.x;
    const int last = np > 0 ? np - 1 : 0;
#pragma unroll
    for (int k = 0; k < SHARE_K; ++k) {
        const int i = t + k * BLOCK;
        v[k] = p[i < last ? i : last];
    }
}
template <int BLOCK, int SHARE_K>
__device__ __forceinline__ double strided_share_sum(const double (&v)[SHARE_K], int np)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < SHARE_K; ++k)
        if ((int)threadIdx.x + k * BLOCK < np) s += v[k];
    return s;
}
template <int BLOCK>
__device__ __forceinline__ double block_sum_all_share(double s, double *lds)
{
    const double r = block_sum<BLOCK>(s, lds);
    if (threadIdx.x == 0) lds[BLOCK / WAVE] = r;
    __syncthreads();
    const double out = lds[BLOCK / WAVE];
    __syncthreads();
    return out;
}

// Blue's range-safe sum of squares (constants of LAPACK 3.10's dnrm2); see scalar.h "range-safe norms".
constexpr double BLUE_TSML = 0x1p-511, BLUE_TBIG = 0x1p486, BLUE_SSML = 0x1p537, BLUE_SBIG = 0x1p-538;

struct Blue3 {
    double sml, med, big;
};
__device__ __forceinline__ void blue_add(Blue3 &a, double x)
{
    const double ax = fabs(x);
    if (ax > BLUE_TBIG) {
        const double t = ax * BLUE_SBIG;
        a.big += t * t;
    } else if (ax < BLUE_TSML) {
        const double t = ax * BLUE_SSML;
        a.sml += t * t;
    } else {
        a.med += ax * ax;  // (NaN lands here)
    }
}
__device__ __forceinline__ double blue_norm(double asml, double amed, double abig)
{
    if (abig > 0.0) {
        if (amed > 0.0 || amed != amed) abig += (amed * BLUE_SBIG) * BLUE_SBIG;
        return sqrt(abig) / BLUE_SBIG;
    }
    if (asml > 0.0) {
        if (amed > 0.0 || amed != amed) {
            const double a = sqrt(amed), b = sqrt(asml) / BLUE_SSML;
            const double ymin = a < b ? a : b, ymax = a < b ? b : a;
            const double q = ymin / ymax;
            return ymax * sqrt(1.0 + q * q);
        }
        return sqrt(asml) / BLUE_SSML;
    }
    return sqrt(amed);
}

// Blocks b and b+8 share an XCD (and its 4 MiB L2) under the observed round-robin
// placement.  Give each XCD label one contiguous eighth of the work items so that
// neighbouring row blocks -- which gather neighbouring parts of x -- share an L2.
// Placement only changes speed, never results.
struct XcdRange {
    int64_t first, end, stride;
};
__device__ __forceinline__ XcdRange xcd_range(int64_t nitems, int g, int wg)
{
    XcdRange r;
    if ((g & 7) != 0 || g < 8) {
        r.first = wg;
        r.end = nitems;
        r.stride = g;
        return r;
    }
    const int xcd = wg & 7, slot = wg >> 3;
    const int64_t per = (nitems + 7) >> 3;
    r.first = (int64_t)xcd * per + slot;
    r.end = (int64_t)(xcd + 1) * per;
    if (r.end > nitems) r.end = nitems;
    r.stride = g >> 3;
    return r;
}

}  // namespace lsqrhip
