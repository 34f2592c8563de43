// csb.h -- "column-swept row blocks": the product for matrices whose columns are scattered over an
// x that does not fit an XCD's L2 (aprod mode 1 / mode 2 on random and power-law systems:
// BASELINE configs 3-5).
//
// Same contract as spmv.h's k_spmv_fused (reference src/lsqr.f90:166-174 / :186-194 fused with the
// dscal before and the dnrm2 after, :681-683 / :692-695):
//
//     y_i  <-  cy * (y_i * sy)  +  sum_j A_ij * (x_j * sx)        partial += (y_i * ns)^2
//
// Why.  With scattered columns every gathered x_j is an L1 miss, and a CU retires only ~0.3 misses
// per clock from L2 (64 outstanding lines / ~220 cycles) and a third of that from beyond
// (scripts/gather_roof.hip).  Column PANELS (spmv.h) bring the gathers into L2 but pay for it with
// row pointers per (row, panel), per-panel row sums Z written and re-read, and a combine launch:
// 1.6x the algorithmic bytes at config 4, 18.6 % of the HBM roofline.  Here instead:
//
//   * rows are cut into blocks of R <= 10112 rows, and the nonzeros of a block are stored SORTED BY
//     COLUMN: one 1024-thread workgroup (one per CU) sweeps x from left to right while it streams
//     its block -- 8-byte value + 4-byte (local row | local column) = 12 bytes per nonzero, the
//     algorithmic minimum, and no row pointers at all.  All workgroups sweep at the same pace, so
//     the part of x they are gathering from is in L2 (scripts/csb_roof.hip: the sweep sustains the
//     L2-resident gather rate with an x of any size, and the 64 lanes of a gather touch
//     neighbouring lines: 176 / 215 / 476 G nonzeros/s at 0.08 / 0.1 / 0.8 nonzeros per column and block).
//   * the block's row sums are accumulated IN LDS with ds_add_f64 (measured free beside the
//     gathers).  Floating-point adds in an order nobody controls would not be reproducible, so each
//     product is first split EXACTLY into two parts on fixed binary grids,
//         hi = the multiple of q0 nearest p,     lo = the multiple of q1 nearest p - hi,
//     with q0 = 2^(E+H-53), q1 = q0 * 2^(H-54): 2^E bounds |p| (max|a| * max|x sx|), 2^(H-1) bounds
//     the nonzeros of a row.  Sums of such multiples stay below 2^53 grid steps, so every ds_add_f64
//     is exact and the result does not depend on the order of the adds -- nor on R, the grid or
//     which workgroup took which block: bit-reproducible by construction.  What is dropped is
//     below q1/2 = 2^(E+2H-108) per product: with rows of <= 2^14 nonzeros 80 bits below the
//     largest possible product, i.e. the row sum is (far) more accurate than the reference's
//     left-to-right sum, and agrees with it to rounding.
//   * the epilogue of a block forms y_i, the block's partial of sum (y ns)^2 (one per BLOCK, so the
//     fixed-order reduction is independent of the launch shape) and clears the accumulators.
//
// Layout (built once by csb_build below from the COO triplets, stable LSD radix sorts of csr_build.h):
//   block b = rows [rstart[b], rstart[b+1]): at most R rows, cut so that every block holds about the
//   same number of NONZEROS (equal sweeps: workgroups that start together must stay within ~1 % of
//   each other's column for the x they gather to be in L2 -- with equal ROW counts a square random
//   matrix's transpose, whose rows are Poisson(100) long, ran 16 % slower than the matrix itself);
//   its nonzeros sorted by column (ties: COO order), padded to whole chunks of 256 (pad = value 0
//   aimed at a dummy accumulator);
//   cptr[b] = first chunk of block b; val[k], idx[k] = lrow << 18 | (col - cbase[k / 256]);
//   cbase[c] = column of the first nonzero of chunk c.  A chunk spans < 2^18 columns or the build
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
constexpr int CSB_RMAX = 10112;                  // rows per block: 2 accumulators of 8 bytes in 160 KB of LDS
constexpr int CSB_RMAX32 = 13504;                // ... with the low parts as 32-bit integers (12 bytes per row)
constexpr int CSB_LO32_MAXH = 10;                // that form is used for rows of <= 512 nonzeros
constexpr int CSB_LDS_BYTES = (CSB_RMAX + 64) * 16;
static_assert((CSB_RMAX32 + 64) * 12 <= CSB_LDS_BYTES, "both accumulator forms share one LDS array");
constexpr int CSB_LCOL_BITS = 18;
constexpr unsigned CSB_LCOL_MASK = (1u << CSB_LCOL_BITS) - 1u;
constexpr int CSB_GRID = 256;                    // one workgroup per CU
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
    int ea;  // 2^ea > max|a_ij|
    int H;   // 2^(H-1) >= nonzeros of the longest row, H >= 3
    int b0, b1;  // the row blocks of THIS launch: [b0, b1)
    int S;       // column splits: S workgroups share a row block, each sweeping 1/S of its chunks (see below)
    double *zhi, *zlo;  // S > 1: the splits' exact partial sums, [S][rows] each
    double *q1out;      // S > 1: this launch's q1 for k_csb_combine
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

// pos1[i] = original position of the i-th nonzero in column order; cnt[row] += 1
__global__ __launch_bounds__(256) void k_csb_pos(const unsigned long long *__restrict__ sorted, int64_t nnz,
                                                 const int *__restrict__ rowk, unsigned *__restrict__ pos1,
                                                 int *__restrict__ cnt)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const unsigned p = (unsigned)(sorted[i] & 0xffffffffull);
        pos1[i] = p;
        atomicAdd(&cnt[rowk[p] - 1], 1);
    }
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
// of the block in column order, or padding.  flags[3] |= 1 if a chunk spans 2^18 columns or more.
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
    const double *xmax;  // partials of max|x| (vec.h k_amax), or null: |x sx| <= 1 is known
    int nxmax;
};

// LO32 = false: hi and lo parts both as doubles on the grids q0, q1 of the header (any row length).
// LO32 = true (rows of <= 512 nonzeros, H <= 10): the low part as a 32-bit INTEGER count of steps of
// q1' = q0 / 2^(31-H) -- |lo| <= q0/2 is at most 2^(30-H) steps, 2^(H-1) of them stay below 2^31 --
// accumulated with ds_add_u32 (integer adds are exact and order-free by nature).  12 bytes per row
// instead of 16: a third more rows per block, i.e. a third more nonzeros per column of x in every sweep
// (config 4: 13021 instead of 9766 rows per block).  Dropped per product: < q1'/2 = 2^(E+2H-85).
template <bool LO32, typename VT = double>
__global__ __launch_bounds__(CSB_BLOCK, 1) void k_spmv_csb(
    CsbMat A, const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef,
    const int *__restrict__ stop, double *__restrict__ partials, const double *__restrict__ pin, int npin,
    const NormSlot *__restrict__ slot_in, NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, CsbX xb,
    NScale nsc)
{
    __shared__ double acc_raw[CSB_LDS_BYTES / 8];
    __shared__ double red[CSB_WAVES + 2];
    constexpr int NR = LO32 ? CSB_RMAX32 + 64 : CSB_RMAX + 64;
    double *const acc_hi = acc_raw;
    double *const acc_lo = acc_raw + NR;                              // LO32 = false
    int *const acc_li = reinterpret_cast<int *>(acc_raw + NR);        // LO32 = true
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

    double sx, sy, cy;
    if (pin != nullptr) {
        // the 256-thread fixed-order reduction of the other kernels, bit for bit
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
        const double nrm = sqrt(red[CSB_WAVES]) * nsc.inv;
        __syncthreads();
        if (skip_if_zero && !(nrm > 0.0)) {  // mode 2 is skipped when beta == 0 (:691)
            if (wg == 0 && tid == 0) {
                slot_out->nrm = nrm;
                slot_out->scale = 1.0;
            }
            return;
        }
        sx = nrm > 0.0 ? 1.0 / nrm : 1.0;
        cy = -nrm;
        sy = slot_in->scale;
        if (wg == 0 && tid == 0) {
            slot_out->nrm = nrm;
            slot_out->scale = sx;
        }
    } else {
        if (coef->skip != 0) return;
        sx = coef->sx;
        sy = coef->sy;
        cy = coef->cy;
    }

    // the binary grids of this launch: 2^E bounds |a_ij * (x_j sx)|
    int ex = 1;  // |x sx| <= 1 (+ rounding) for the solver's own unit vectors
    if (xb.xmax != nullptr) {
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
        ex = 0;
        if (bound > 0.0 && bound < 1.0e308) (void)frexp(bound, &ex);  // bound < 2^ex
    }
    int E = A.ea + ex;
    E = E > 1020 - A.H ? 1020 - A.H : E;                        // C0 must stay finite
    E = E < 108 - 2 * A.H - 1020 ? 108 - 2 * A.H - 1020 : E;    // q1 must stay normal
    const double C0 = ldexp(1.5, E + A.H - 1);        // 1.5 * 2^52 * q0
    const int e1 = LO32 ? E + 2 * A.H - 84 : E + 2 * A.H - 107;   // q1 = 2^e1
    const double C1 = ldexp(1.5, e1 + 52);            // 1.5 * 2^52 * q1
    const long long C1bits = __double_as_longlong(C1);
    const double q1 = ldexp(1.0, e1);
    const double pmax = ldexp(1.0, E);

    for (int i = tid; i < NR; i += CSB_BLOCK) {
        acc_hi[i] = 0.0;
        if (LO32) acc_li[i] = 0;
        else acc_lo[i] = 0.0;
    }
    __syncthreads();

    if (A.S > 1 && wg == 0 && tid == 0) *A.q1out = q1;
    // Column splits (few rows: fewer row blocks than CUs).  S workgroups share a block, each sweeping a
    // contiguous S-th of its column-sorted chunks into accumulators of its own; their hi / lo sums go to
    // zhi / zlo and k_csb_combine adds them -- sums on the grids are exact, so the result is bit for bit
    // what ONE workgroup would have produced.  A block of R rows then still holds R d / n nonzeros per
    // column although 256 / S blocks cover the matrix: R can stay large (one rank's block of config 4 at
    // N = 8: 9766 instead of 4883 rows per block).
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
                double hi = (p + C0) - C0;
                const double t1 = (p - hi) + C1;       // C1 + (steps of q1): the steps sit in the low mantissa bits
                const bool out = !(fabs(p) <= pmax);   // beyond the bound (or not finite): added as it is
                if (out) hi = p;
                atomicAdd(&acc_hi[r], hi);
                if (LO32) {
                    const int steps = out ? 0 : (int)(__double_as_longlong(t1) - C1bits);
                    atomicAdd(&acc_li[r], steps);
                } else {
                    atomicAdd(&acc_lo[r], out ? 0.0 : t1 - C1);
                }
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
        __syncthreads();
        // epilogue of the block: y, its partial of sum (y ns)^2, accumulators cleared
        const int row0 = A.rstart[b];
        const int nr = A.rstart[b + 1] - row0;
        double sq = 0.0;
        if (A.S > 1) {  // a split: the exact sums as they are (integer steps of q1 as doubles when LO32)
            double *zh = A.zhi + (size_t)sp * A.rows + row0, *zl = A.zlo + (size_t)sp * A.rows + row0;
            for (int r = tid; r < nr; r += CSB_BLOCK) {
                zh[r] = acc_hi[r];
                zl[r] = LO32 ? (double)acc_li[r] : acc_lo[r];
                acc_hi[r] = 0.0;
                if (LO32) acc_li[r] = 0;
                else acc_lo[r] = 0.0;
            }
            if (tid == 0) {
                acc_hi[A.R] = 0.0;
                if (LO32) acc_li[A.R] = 0;
                else acc_lo[A.R] = 0.0;
            }
            __syncthreads();
            continue;
        }
        for (int r = tid; r < nr; r += CSB_BLOCK) {
            const double hi = acc_hi[r];
            const double lo = LO32 ? (double)acc_li[r] * q1 : acc_lo[r];
            acc_hi[r] = 0.0;
            if (LO32) acc_li[r] = 0;
            else acc_lo[r] = 0.0;
            const VT yn = (VT)(cy * ((double)y[row0 + r] * sy) + (hi + lo));
            y[row0 + r] = yn;
            const double ys = (double)yn * nsc.s;
            sq += ys * ys;
        }
        if (tid == 0) {  // the padding's dummy accumulator
            acc_hi[A.R] = 0.0;
            if (LO32) acc_li[A.R] = 0;
            else acc_lo[A.R] = 0.0;
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
template <bool LO32, typename VT = double>
__global__ __launch_bounds__(CSB_BLOCK) void k_csb_combine(
    CsbMat A, VT *__restrict__ y, const SpmvCoef *__restrict__ coef, const int *__restrict__ stop,
    double *__restrict__ partials, const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in,
    int skip_if_zero, NScale nsc)
{
    __shared__ double red[CSB_WAVES + 2];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (*stop != 0) return;
    double sy, cy;
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
        const double nrm = sqrt(red[CSB_WAVES]) * nsc.inv;
        __syncthreads();
        if (skip_if_zero && !(nrm > 0.0)) return;
        cy = -nrm;
        sy = slot_in->scale;
    } else {
        if (coef->skip != 0) return;
        sy = coef->sy;
        cy = coef->cy;
    }
    const double q1 = *A.q1out;
    for (int b = blockIdx.x; b < A.nrb; b += gridDim.x) {
        const int row0 = A.rstart[b];
        const int nr = A.rstart[b + 1] - row0;
        double sq = 0.0;
        for (int r = tid; r < nr; r += CSB_BLOCK) {
            double hi = A.zhi[row0 + r], lo = A.zlo[row0 + r];
            for (int sp = 1; sp < A.S; ++sp) {
                hi = hi + A.zhi[(size_t)sp * A.rows + row0 + r];
                lo = lo + A.zlo[(size_t)sp * A.rows + row0 + r];
            }
            if (LO32) lo = lo * q1;
            const VT yn = (VT)(cy * ((double)y[row0 + r] * sy) + (hi + lo));
            y[row0 + r] = yn;
            const double ys = (double)yn * nsc.s;
            sq += ys * ys;
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
