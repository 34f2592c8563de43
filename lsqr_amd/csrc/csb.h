// csb.h -- "column-swept row blocks": the product for matrices whose columns are scattered over an
// x that does not fit an XCD's L2 (aprod mode 1 / mode 2 on random and power-law systems:
// BASELINE configs 3-5).
//
// Same contract as spmv.h's k_spmv_fused (reference src/lsqr.f90:166-174 / :186-194 fused with the
// dscal before and the dnrm2 after, :681-683 / :692-695):
//
//     y_i  <-  cy * (y_i * sy)  +  sum_j A_ij * (x_j * sx)        partial += (y_i * ns)^2
//
// Why.  With scattered columns every gathered x_j is an L1 miss, and a CU keeps only ~64 cache lines in
// flight: ~0.3 misses per clock from L2 (~220 cycles), a third of that from beyond (scripts/gather_roof.hip),
// and the (value, index) stream from HBM competes for the same slots.  Column PANELS (spmv.h) bring the
// gathers into L2 but pay for it with row pointers per (row, panel), per-panel row sums Z written and
// re-read, and a combine launch: 1.6x the algorithmic bytes at config 4.  Here instead:
//
//   * rows are cut into blocks of R <= 20352 rows, and the nonzeros of a block are stored SORTED BY
//     COLUMN: one 1024-thread workgroup (one per CU) sweeps x from left to right while it streams
//     its block -- 8-byte value + 4-byte (local row | local column) = 12 bytes per nonzero, the
//     algorithmic minimum, and no row pointers at all.  All workgroups sweep at the same pace, so
//     the part of x they are gathering from is in L2, and the 64 lanes of a gather touch neighbouring
//     lines: what a product costs is the number of LINES of x a block touches, i.e. it falls with
//     R d / n, the nonzeros a block holds per column -- hence R as large as the LDS allows.
//   * the block's row sums are accumulated IN LDS, one 64-bit INTEGER per row (ds_add_u64).  Floating-
//     point adds in an order nobody controls would not be reproducible; integer adds are exact and
//     order-free by nature.  Each product p is rounded once to the fixed binary grid
//         g = 2^(eb - 61),      2^eb > B >= |sum_j A_ij (x_j sx)| for every row i,
//     q = rint(p / g) is added, and y's new part is (double)(sum of the q) * g: ONE more rounding.
//     The bound: B = max_i sum_j |a_ij| * max_j |x_j sx| -- the largest row 1-norm, taken at build time as an
//     integer sum (deterministic), times the largest entry of the vector, taken by a k_amax pass over x before
//     every product (0.4 % of a config-4 product).  It holds whatever the matrix looks like -- in particular with
//     DUPLICATE (i, j) entries, which the reference sums like any others: the first form of this bound,
//     |a_i|_2 |x|_2 for the solver's unit vectors, is wrong with duplicates (185 equal entries on one (i, j)
//     weigh 185, not sqrt(185): the sums wrapped on such a system, tests/test_gpu_fuzz.py) -- and for vectors
//     without outliers it is also the tighter one (config 4: 60 * 1.6e-3 against 6.5).  Only the FINAL sum has to
//     fit -- two's-complement adds wrap, so partial sums in any order may overflow on the way -- and it does
//     with two bits to spare whatever the row length.  So the result does not depend on the order of the adds,
//     on R, on the launch shape, on which workgroup took which block or on column splits: bit-reproducible by
//     construction.  Accuracy: each product is off by <= g/2 = 2^-62 B, i.e. a row of k nonzeros by <= k 2^-62 B
//     (typically sqrt(k) 2^-63 B) where the reference's left-to-right sum is off by up to k 2^-53 |a_i|'|x|.
//     (r02 kept two parts per row, 12-16 bytes: 9766-13021 rows per block at config 4; 8 bytes per
//     row give 19532 -- 50 % more nonzeros per column of x in every sweep and 2 rounds instead of 3.)
//   * a product beyond the bound or not finite (inf / NaN in x: never in a solve that has not
//     already failed) cannot enter an integer sum: the sweep leaves it out and raises a flag, and
//     the block's epilogue then reads the stream a second time, adds ONLY those products as doubles
//     and patches the rows concerned -- inf and NaN come out as IEEE addition gives them, like the
//     reference's (tests/test_gpu_csb.py::test_non_finite_and_huge_x_as_the_reference).
//   * the epilogue of a block forms y_i, the block's partial of sum (y ns)^2 (one per BLOCK, so the
//     fixed-order reduction is independent of the launch shape) and clears the accumulators.
//
// Layout (built once by build_csb from the COO triplets, stable LSD radix sorts of csr_build.h):
//   block b = rows [rstart[b], rstart[b+1]): at most R rows, cut so that every block holds about the
//   same number of NONZEROS (equal sweeps: workgroups that start together must stay within ~1 % of
//   each other's column for the x they gather to be in L2 -- with equal ROW counts a square random
//   matrix's transpose, whose rows are Poisson(100) long, ran 16 % slower than the matrix itself);
//   its nonzeros sorted by column (ties: COO order), padded to whole chunks of 256 (pad = value 0
//   aimed at a dummy accumulator);
//   cptr[b] = first chunk of block b; val[k], idx[k] = lrow << 17 | (col - cbase[k / 256]);
//   cbase[c] = column of the first nonzero of chunk c.  A chunk spans < 2^17 columns or the build
//   gives up (an almost empty block: such a matrix keeps the panel layout).
#pragma once

#include "common.h"
#include "csr_build.h"
#include "scalar.h"
#include "state.h"

namespace lsqrhip {

constexpr int CSB_BLOCK = 1024;
constexpr int CSB_WAVES = CSB_BLOCK / WAVE;
constexpr int CSB_U = 4;                         // nonzeros per lane and step
constexpr int CSB_CHUNK = CSB_U * WAVE;          // 256: what one wave takes per step
constexpr int CSB_RMAX = 20352;                  // rows per block: one 8-byte accumulator each in 160 KB of LDS
constexpr int CSB_NACC = CSB_RMAX + 64;          // + the padding's dummy accumulator (index R)
static_assert(CSB_NACC * 8 + (CSB_WAVES + 2) * 8 + 16 <= 160 * 1024, "accumulators + reduction scratch fit the LDS");
constexpr int CSB_LCOL_BITS = 17;
constexpr unsigned CSB_LCOL_MASK = (1u << CSB_LCOL_BITS) - 1u;
static_assert(CSB_NACC <= (1 << (32 - CSB_LCOL_BITS)), "local rows fit the index word");
constexpr int CSB_GRID = 256;                    // one workgroup per CU
constexpr int CSB_NORM_FRAC = 32;                // build: row 1-norms as integer sums of ceil(|a| 2^(32 - ea))
// The (value, index) stream is read once: loaded non-temporal so that it does not push the part of x the
// XCD's workgroups are gathering from out of L2 (PMC before: 15 % of the gathers missed L2, 2.6x the
// layout's bytes fetched; config 4 4.80 -> 4.20 ms, config 3 at 100 per row 956 -> 900 us).
#ifndef CSB_NT_STREAM
#define CSB_NT_STREAM 1
#endif

struct CsbMat {
    const void *val;        // VT values (double; float for a REAL32 handle)
    const unsigned *idx;
    const int *cbase;
    const long long *cptr;  // [nrb + 1], in chunks
    const int *rstart;      // [nrb + 1] first row of each block (blocks are cut by NONZEROS, at most R rows each)
    int nrb, R, rows, cols; // R = the dummy accumulator's index = rows per block at most
    int e1;  // 2^e1 > max_i sum_j |a_ij|: with max|x sx| the bound on a row sum
    int b0, b1;  // the row blocks of THIS launch: [b0, b1)
    int S;       // column splits: S workgroups share a row block, each sweeping 1/S of its chunks (see below)
    long long *z;   // S > 1: the splits' exact integer sums, [S][rows]
    int *bad;       // S > 1: [nrb] a split of the block left a product out (beyond the bound / not finite)
    double *gout;   // S > 1: this product's grid step g for k_csb_combine
};

// ---------------------------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------------------------
// packed[i] = (col-1) << 32 | i;  flags[0] |= bad row, flags[2] |= bad column, flags[1] |= not sorted by column
__global__ __launch_bounds__(256) void k_csb_pack_col(const int *__restrict__ rowk, const int *__restrict__ colk,
                                                      int64_t nnz, int rows, int cols,
                                                      unsigned long long *__restrict__ packed, int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int badr = 0, badc = 0, uns = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const int r = rowk[i];
        int c = colk[i];
        if (r < 1 || r > rows) badr = 1;
        if (c < 1 || c > cols) { badc = 1; c = 1; }
        if (i > 0 && colk[i - 1] > c) uns = 1;
        packed[i] = ((unsigned long long)(unsigned)(c - 1) << 32) | (unsigned long long)(unsigned)i;
    }
    if (badr) atomicOr(&flags[0], 1);
    if (badc) atomicOr(&flags[2], 1);
    if (uns) atomicOr(&flags[1], 1);
}

// pos1[i] = original position of the i-th nonzero in column order; cnt[row] += 1; and the row 1-norms behind
// the bound on a row sum (header): n1[row] += ceil(|a| 2^(32 - ea)) with 2^ea > max|a| -- an integer sum, so the
// bound does not depend on the order of the atomics.  (A value that is not finite counts as 2^ea: its products
// are left to the outlier pass anyway.)
__global__ __launch_bounds__(256) void k_csb_pos(const unsigned long long *__restrict__ sorted, int64_t nnz,
                                                 const int *__restrict__ rowk, const double *__restrict__ a, double sc,
                                                 unsigned *__restrict__ pos1, int *__restrict__ cnt,
                                                 unsigned long long *__restrict__ n1)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double one = (double)(1ull << CSB_NORM_FRAC);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const unsigned p = (unsigned)(sorted[i] & 0xffffffffull);
        pos1[i] = p;
        const int r = rowk[p] - 1;
        atomicAdd(&cnt[r], 1);
        double t = fabs(a[p]) * sc;      // in [0, 1)
        t = t < 1.0 ? t : 1.0;           // (inf, NaN -> 1)
        atomicAdd(&n1[r], (unsigned long long)ceil(t * one));
    }
}

__global__ __launch_bounds__(256) void k_csb_maxu64(const unsigned long long *__restrict__ a, int64_t n,
                                                    unsigned long long *__restrict__ out)
{
    unsigned long long m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = a[i] > m ? a[i] : m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(m, off, WAVE);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && m > 0) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void k_csb_maxint(const int *__restrict__ a, int64_t n, int *__restrict__ out)
{
    int m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = max(m, a[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, WAVE));
    if ((threadIdx.x & (WAVE - 1)) == 0 && m > 0) atomicMax(out, m);
}

// packed[i] = block(row of the i-th nonzero in column order) << 32 | i;  block b = rows [rstart[b], rstart[b+1])
__global__ __launch_bounds__(256) void k_csb_pack_rb(const int *__restrict__ rowk, const unsigned *__restrict__ pos1,
                                                     int64_t nnz, const int *__restrict__ rstart, int nrb,
                                                     unsigned long long *__restrict__ packed)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const int r = rowk[pos1[i]] - 1;
        int lo = 0, hi = nrb - 1;  // last b with rstart[b] <= r
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rstart[mid] <= r) lo = mid;
            else hi = mid - 1;
        }
        packed[i] = ((unsigned long long)(unsigned)lo << 32) | (unsigned long long)(unsigned)i;
    }
}

// One workgroup per chunk: element t of chunk c of block b is the (c - cptr[b]) * 256 + t -th nonzero
// of the block in column order, or padding.  flags[3] |= 1 if a chunk spans 2^17 columns or more.
__global__ __launch_bounds__(CSB_CHUNK) void k_csb_fill(const unsigned long long *__restrict__ sorted2,
                                                        const unsigned *__restrict__ pos1,
                                                        const int *__restrict__ rowk, const int *__restrict__ colk,
                                                        const double *__restrict__ a,
                                                        const long long *__restrict__ rbstart,
                                                        const long long *__restrict__ cptr,
                                                        const int *__restrict__ rstart, int nrb, int R,
                                                        double *__restrict__ val, unsigned *__restrict__ idx,
                                                        int *__restrict__ cbase, int *__restrict__ flags)
{
    __shared__ int s_b, s_cb;
    const long long c = blockIdx.x;
    if (threadIdx.x == 0) {
        int lo = 0, hi = nrb - 1;  // last b with cptr[b] <= c
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cptr[mid] <= c) lo = mid;
            else hi = mid - 1;
        }
        s_b = lo;
    }
    __syncthreads();
    const int b = s_b;
    const long long e = (c - cptr[b]) * CSB_CHUNK + threadIdx.x;  // rank inside the block
    const long long j0 = rbstart[b], j1 = rbstart[b + 1];
    const bool real = j0 + e < j1;
    int col = 0, lrow = R;  // padding: the dummy accumulator
    double v = 0.0;
    if (real) {
        const unsigned i = (unsigned)(sorted2[j0 + e] & 0xffffffffull);
        const unsigned p = pos1[i];
        col = colk[p] - 1;
        lrow = (rowk[p] - 1) - rstart[b];
        v = a[p];
    }
    if (threadIdx.x == 0) s_cb = col;  // the first element of a chunk is never padding
    __syncthreads();
    const int cb = s_cb;
    const int lc = real ? col - cb : 0;
    if (lc < 0 || lc > (int)CSB_LCOL_MASK) atomicOr(&flags[3], 1);
    const long long k = c * CSB_CHUNK + threadIdx.x;
    val[k] = v;
    idx[k] = ((unsigned)lrow << CSB_LCOL_BITS) | ((unsigned)lc & CSB_LCOL_MASK);
    if (threadIdx.x == 0) cbase[c] = cb;
}

// ---------------------------------------------------------------------------------------------
// product
// ---------------------------------------------------------------------------------------------
struct CsbX {
    const double *xmax;  // partials of max|x| (vec.h k_amax over the vector this product gathers from)
    int nxmax;
};

// sx, sy, cy of this launch: explicit (coef) or lazy from the previous kernel's partials (pin) -- the
// 256-thread fixed-order reduction of the other kernels, bit for bit.  `nrm` is the lazy norm.
struct CsbCoef {
    double sx, sy, cy, nrm;
    bool skip;
};
__device__ __forceinline__ CsbCoef csb_coef(const SpmvCoef *__restrict__ coef, const double *__restrict__ pin, int npin,
                                            const NormSlot *__restrict__ slot_in, int skip_if_zero, NScale nsc,
                                            double *red)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    CsbCoef c{1.0, 1.0, 1.0, 0.0, false};
    if (pin != nullptr) {
        double s = 0.0;
        if (tid < SC_BLOCK && npin > 0) s = strided_sum<SC_BLOCK>(pin, npin);
        s = wave_sum(s);
        if (tid < SC_BLOCK && lane == 0) red[w] = s;
        __syncthreads();
        if (tid == 0) {
            double r = 0.0;
#pragma unroll
            for (int i = 0; i < SC_BLOCK / WAVE; ++i) r += red[i];
            red[CSB_WAVES] = r;
        }
        __syncthreads();
        c.nrm = sqrt(red[CSB_WAVES]) * nsc.inv;
        __syncthreads();
        c.skip = skip_if_zero && !(c.nrm > 0.0);   // mode 2 is skipped when beta == 0 (:691)
        c.sx = c.nrm > 0.0 ? 1.0 / c.nrm : 1.0;
        c.cy = -c.nrm;
        c.sy = slot_in->scale;
    } else {
        c.skip = coef->skip != 0;
        c.sx = coef->sx;
        c.sy = coef->sy;
        c.cy = coef->cy;
    }
    return c;
}

// 2^eb > the bound on |row sum| of this launch (header): the largest row 1-norm times max|x sx|, the latter from
// the partials of the k_amax pass that precedes every product; every thread gets the same value.
__device__ __forceinline__ int csb_bound_exp(const CsbMat &A, CsbX xb, double sx, double *red)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double m = 0.0;
    for (int i = tid; i < xb.nxmax; i += CSB_BLOCK) m = fmax(m, xb.xmax[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, WAVE));
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < CSB_WAVES; ++i) m = fmax(m, red[i]);
    __syncthreads();
    const double bound = m * fabs(sx);
    int ex = 0;
    if (bound > 0.0 && bound < 1.0e308) (void)frexp(bound, &ex);  // bound < 2^ex
    int eb = A.e1 + ex;
    eb = eb > 1020 ? 1020 : eb;      // 2^eb and 2^(61 - eb) must stay finite and normal
    eb = eb < -960 ? -960 : eb;
    return eb;
}

// Rows that were sent a product beyond the bound or not finite.  The sweep left those products out (an
// integer sum cannot hold them); here the chunks [c0, c1) of the block are read once more, ONLY those
// products are added -- as doubles, in LDS (`accd`: the block's accumulators, all zero on entry and on
// exit) -- and the rows concerned are patched in y: inf and NaN come out as IEEE addition gives them,
// which is what the reference's row sum does with them.  Reached when x (or A) holds inf / NaN or
// |x| is not what the bound was taken from; never by a healthy solve.
template <typename VT>
__device__ void csb_add_outliers(const CsbMat &A, const VT *__restrict__ aval, long long c0, long long c1,
                                 const VT *__restrict__ x, double sx, double pmax, double *accd,
                                 VT *__restrict__ y, int row0, int nr)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (long long c = c0 + w; c < c1; c += CSB_WAVES) {
        const int base = A.cbase[c];
        const long long k = c * CSB_CHUNK + lane;
#pragma unroll
        for (int j = 0; j < CSB_U; ++j) {
            const unsigned i = A.idx[k + j * WAVE];
            const double p = (double)aval[k + j * WAVE] * ((double)x[base + (int)(i & CSB_LCOL_MASK)] * sx);
            const int r = (int)(i >> CSB_LCOL_BITS);
            if (!(fabs(p) < pmax) && r < nr) atomicAdd(&accd[r], p);
        }
    }
    __syncthreads();
    for (int r = tid; r < nr; r += CSB_BLOCK) {
        const double v = accd[r];
        if (v != 0.0) {   // (true for NaN)
            y[row0 + r] = (VT)((double)y[row0 + r] + v);
            accd[r] = 0.0;
        }
    }
    __syncthreads();
}

// the block's partial of sum (y ns)^2 from y as stored, with the epilogue's thread -> row mapping and reduction
template <typename VT>
__device__ __forceinline__ double csb_sumsq_rows(const VT *__restrict__ y, int row0, int nr, NScale nsc)
{
    double sq = 0.0;
    for (int r = threadIdx.x; r < nr; r += CSB_BLOCK) {
        const double ys = (double)y[row0 + r] * nsc.s;
        sq += ys * ys;
    }
    return sq;
}

template <typename VT = double>
__global__ __launch_bounds__(CSB_BLOCK, 1) void k_spmv_csb(
    CsbMat A, const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef,
    const int *__restrict__ stop, double *__restrict__ partials, const double *__restrict__ pin, int npin,
    const NormSlot *__restrict__ slot_in, NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, CsbX xb,
    NScale nsc)
{
    __shared__ unsigned long long acc[CSB_NACC];
    __shared__ double red[CSB_WAVES + 2];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {  // the scalar rider is written for 256 threads: the other waves leave
        if (tid >= SC_BLOCK) return;
        run_rider(rider, red);
        return;
    }
    if (*stop != 0) return;
    const VT *__restrict__ aval = static_cast<const VT *>(A.val);

    const CsbCoef co = csb_coef(coef, pin, npin, slot_in, skip_if_zero, nsc, red);
    if (pin != nullptr && wg == 0 && tid == 0) {
        slot_out->nrm = co.nrm;
        slot_out->scale = co.skip ? 1.0 : co.sx;
    }
    if (co.skip) return;
    const double sx = co.sx, sy = co.sy, cy = co.cy;

    // the binary grid of this launch
    const int eb = csb_bound_exp(A, xb, sx, red);
    const double pmax = ldexp(1.0, eb);          // a product is in range below this
    const double ginv = ldexp(1.0, 61 - eb);     // 1 / g
    const double g = ldexp(1.0, eb - 61);

    for (int i = tid; i < CSB_NACC; i += CSB_BLOCK) acc[i] = 0ull;
    if (tid == 0) s_bad = 0;
    __syncthreads();

    if (A.S > 1 && wg == 0 && tid == 0) *A.gout = g;
    // Column splits (few rows: fewer row blocks than CUs).  S workgroups share a block, each sweeping a
    // contiguous S-th of its column-sorted chunks into accumulators of its own; their integer sums go to
    // z and k_csb_combine adds them -- exact, so the result is bit for bit what ONE workgroup would have
    // produced.  A block of R rows then still holds R d / n nonzeros per column although 256 / S blocks
    // cover the matrix: R can stay large (one rank's block of config 4 at N = 8).
    const int nunits = (A.b1 - A.b0) * A.S;
    for (int u = wg; u < nunits; u += nwg) {
        const int b = A.b0 + u / A.S, sp = u % A.S;
        const long long cb0 = A.cptr[b], cb1 = A.cptr[b + 1];
        const long long c0 = cb0 + ((cb1 - cb0) * sp) / A.S, c1 = cb0 + ((cb1 - cb0) * (sp + 1)) / A.S;
        // software pipeline: the (value, index) stream of the wave's NEXT chunk is in flight while the
        // gathers and the LDS adds of this one run (two register sets, loads unconditional: clamped)
        double av[CSB_U], bv[CSB_U];
        unsigned iv[CSB_U], jv[CSB_U];
        int cb = 0, cbn = 0;
        bool outlier = false;
        const long long clast = c1 > c0 ? c1 - 1 : c0;
        auto issue = [&](long long c, double (&a)[CSB_U], unsigned (&i)[CSB_U], int &base) {
            const long long cc = c < clast ? c : clast;
            base = A.cbase[cc];
            const long long k = cc * CSB_CHUNK + lane;
#pragma unroll
            for (int j = 0; j < CSB_U; ++j) {
                if (CSB_NT_STREAM) {   // read-once stream: non-temporal, so that it does not push x out of L2
                    a[j] = (double)__builtin_nontemporal_load(&aval[k + j * WAVE]);
                    i[j] = __builtin_nontemporal_load(&A.idx[k + j * WAVE]);
                } else {
                    a[j] = (double)aval[k + j * WAVE];
                    i[j] = A.idx[k + j * WAVE];
                }
            }
        };
        auto work = [&](const double (&a)[CSB_U], const unsigned (&i)[CSB_U], int base) {
            double xv[CSB_U];
#pragma unroll
            for (int j = 0; j < CSB_U; ++j) xv[j] = (double)x[base + (int)(i[j] & CSB_LCOL_MASK)];
#pragma unroll
            for (int j = 0; j < CSB_U; ++j) {
                const double p = a[j] * (xv[j] * sx);
                const int r = (int)(i[j] >> CSB_LCOL_BITS);
                const bool out = !(fabs(p) < pmax);   // beyond the bound (or not finite): left to the outlier pass
                outlier |= out;
                const long long q = __double2ll_rn(p * ginv);
                atomicAdd(&acc[r], out ? 0ull : (unsigned long long)q);
            }
        };
        if (c0 + w < c1) {
            issue(c0 + w, av, iv, cb);
            for (long long c = c0 + w; c < c1; c += 2 * CSB_WAVES) {
                issue(c + CSB_WAVES, bv, jv, cbn);
                work(av, iv, cb);
                if (c + CSB_WAVES < c1) {  // uniform
                    issue(c + 2 * CSB_WAVES, av, iv, cb);
                    work(bv, jv, cbn);
                }
            }
        }
        if (outlier) s_bad = 1;
        __syncthreads();
        // epilogue of the block: y, its partial of sum (y ns)^2, accumulators cleared
        const int row0 = A.rstart[b];
        const int nr = A.rstart[b + 1] - row0;
        const bool bad = s_bad != 0;
        __syncthreads();   // (everyone has read the flag: it may be lowered)
        if (A.S > 1) {  // a split: the exact sums as they are
            long long *zs = A.z + (size_t)sp * A.rows + row0;
            for (int r = tid; r < nr; r += CSB_BLOCK) {
                __builtin_nontemporal_store((long long)acc[r], &zs[r]);   // (written once, read once by k_csb_combine)
                acc[r] = 0ull;
            }
            if (tid == 0) {
                acc[A.R] = 0ull;
                s_bad = 0;
                if (bad) atomicOr(&A.bad[b], 1);
            }
            __syncthreads();
            continue;
        }
        double sq = 0.0;
        for (int r = tid; r < nr; r += CSB_BLOCK) {
            const double sum = (double)(long long)acc[r] * g;
            acc[r] = 0ull;
            const VT yn = (VT)(cy * ((double)y[row0 + r] * sy) + sum);
            y[row0 + r] = yn;
            const double ys = (double)yn * nsc.s;
            sq += ys * ys;
        }
        if (tid == 0) {  // the padding's dummy accumulator
            acc[A.R] = 0ull;
            s_bad = 0;
        }
        if (bad) {  // uniform
            __syncthreads();
            csb_add_outliers<VT>(A, aval, c0, c1, x, sx, pmax, reinterpret_cast<double *>(acc), y, row0, nr);
            sq = csb_sumsq_rows<VT>(y, row0, nr, nsc);
        }
        sq = wave_sum(sq);
        if (lane == 0) red[w] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < CSB_WAVES; ++i) t += red[i];
            partials[b] = t;
        }
        __syncthreads();
    }
}

// The second launch of a column-split product: y and the blocks' partials of sum (y ns)^2 from the
// splits' exact sums.  One workgroup per row block with the thread -> row mapping and the reduction of
// k_spmv_csb's own epilogue, so y AND the partials are bit for bit those of the unsplit kernel.
template <typename VT = double>
__global__ __launch_bounds__(CSB_BLOCK) void k_csb_combine(
    CsbMat A, const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef,
    const int *__restrict__ stop, double *__restrict__ partials, const double *__restrict__ pin, int npin,
    const NormSlot *__restrict__ slot_in, int skip_if_zero, NScale nsc)
{
    __shared__ double accd[CSB_NACC];   // the outlier pass only (all zero otherwise)
    __shared__ double red[CSB_WAVES + 2];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (*stop != 0) return;
    const CsbCoef co = csb_coef(coef, pin, npin, slot_in, skip_if_zero, nsc, red);
    if (co.skip) return;
    const double sx = co.sx, sy = co.sy, cy = co.cy;
    const double g = *A.gout;
    const VT *__restrict__ aval = static_cast<const VT *>(A.val);
    bool cleared = false;
    for (int b = blockIdx.x; b < A.nrb; b += gridDim.x) {
        const int row0 = A.rstart[b];
        const int nr = A.rstart[b + 1] - row0;
        double sq = 0.0;
        for (int r = tid; r < nr; r += CSB_BLOCK) {
            long long s = __builtin_nontemporal_load(&A.z[row0 + r]);
            for (int sp = 1; sp < A.S; ++sp) s += __builtin_nontemporal_load(&A.z[(size_t)sp * A.rows + row0 + r]);
            const double sum = (double)s * g;
            const VT yn = (VT)(cy * ((double)y[row0 + r] * sy) + sum);
            y[row0 + r] = yn;
            const double ys = (double)yn * nsc.s;
            sq += ys * ys;
        }
        if (A.bad[b] != 0) {  // uniform: a split of this block left products out (k_spmv_csb "outlier")
            if (!cleared) {
                for (int i = tid; i < CSB_NACC; i += CSB_BLOCK) accd[i] = 0.0;
                cleared = true;
            }
            __syncthreads();   // (y of this block is written; accd is zero)
            csb_add_outliers<VT>(A, aval, A.cptr[b], A.cptr[b + 1], x, sx, g * 0x1p61, accd, y, row0, nr);
            sq = csb_sumsq_rows<VT>(y, row0, nr, nsc);
            if (tid == 0) A.bad[b] = 0;
        }
        sq = wave_sum(sq);
        if (lane == 0) red[w] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < CSB_WAVES; ++i) t += red[i];
            partials[b] = t;
        }
        __syncthreads();
    }
}

}  // namespace lsqrhip
