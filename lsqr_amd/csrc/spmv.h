// spmv.h -- the aprod hot path: fused CSR SpMV for mode 1 (CSR of A) and mode 2
// (CSR of A', i.e. the transposed product is a plain row-wise SpMV too).
//
// Replaces reference src/lsqr.f90:166-174 (mode 1) and :186-194 (mode 2), fused
// with the dscal before it and the dnrm2 after it (src/lsqr.f90:681-683 and
// :692-695):
//
//     y_i  <-  cy * (y_i * sy)  +  sum_j A_ij * (x_j * sx)        partial += y_i^2
//
// With (sx, sy, cy) = (1, 1, 1) it is exactly aprod_ez's `y = y + A x`.
//
// Work decomposition ("row windows").  Rows are cut into row blocks at build
// time by k_row_blocks: with the work coordinate w(r) = rowptr[r] + r, block k
// owns the rows whose w lies in [k*C, (k+1)*C).  Hence a block has at most C
// rows and, apart from a possible long last row (>= C nonzeros), fewer than 2C
// nonzeros, whatever the degree distribution (banded, random, power law).
//   phase 1  the workgroup streams the block's contiguous (val, col) range with
//            fully coalesced loads, gathers x and stages the products in LDS;
//   phase 2  rows are reduced out of LDS by groups of G lanes (G = 1 for short
//            rows: plain left-to-right sums in CSR = COO order, bit-identical to
//            the reference's row sums; G up to 64 with a shuffle tree otherwise);
//   phase 3  a long last row is split across the whole workgroup (strided
//            partial sums, wave shuffle + LDS reduce).
// The grid is capped and XCD-aware (common.h); every workgroup writes exactly one
// partial of sum(y^2), reduced in fixed order by the scalar kernel => deterministic.
//
// Algorithmic HBM bytes per launch:  12*nnz + P*(rows+1) + 8*cols + 16*rows.
#pragma once

#include "common.h"
#include "state.h"

namespace lsqrhip {

constexpr int SPMV_BLOCK = 256;
constexpr int SPMV_C = 1024;          // window size in work units (nonzeros + rows)
constexpr int SPMV_LDS = 2 * SPMV_C;  // products staged per row block (doubles)
constexpr int SPMV_MAX_GRID = 2048;   // 8 workgroups per CU x 256 CUs

// rb[k] = first row r in [0, m] with rowptr[r] + r >= k*C ; rb[nblk] = m.
template <typename OffT>
__global__ void k_row_blocks(const OffT *__restrict__ rowptr, int m, int64_t nblk, int *__restrict__ rb)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nblk) return;
    if (k == nblk) {
        rb[k] = m;
        return;
    }
    const int64_t target = k * (int64_t)SPMV_C;
    int lo = 0, hi = m;  // answer in [lo, hi]
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if ((int64_t)rowptr[mid] + mid >= target) hi = mid;
        else lo = mid + 1;
    }
    rb[k] = lo;
}

template <typename OffT>
__global__ __launch_bounds__(SPMV_BLOCK) void k_spmv_fused(
    const OffT *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val,
    const int *__restrict__ rb, int64_t nblk, const double *__restrict__ x, double *__restrict__ y,
    const SpmvCoef *__restrict__ coef, const int *__restrict__ stop, double *__restrict__ partials)
{
    if (*stop != 0 || coef->skip != 0) return;
    const double sx = coef->sx, sy = coef->sy, cy = coef->cy;

    __shared__ double prod[SPMV_LDS];
    __shared__ double red[SPMV_BLOCK / WAVE];
    const int tid = threadIdx.x;
    double sq = 0.0;  // this thread's share of sum(y_new^2)

    const XcdRange xr = xcd_range(nblk);
    for (int64_t b = xr.first; b < xr.end; b += xr.stride) {
        const int r0 = rb[b], r1 = rb[b + 1];
        if (r0 >= r1) continue;  // uniform
        const OffT p0 = rowptr[r0];
        const OffT plast = rowptr[r1 - 1], pend = rowptr[r1];
        const bool has_long = (pend - plast) >= (OffT)SPMV_C;
        const int r1s = has_long ? r1 - 1 : r1;
        const int cnt = (int)((has_long ? plast : pend) - p0);  // < 2C by construction

        // ---- phase 1: stream (val, col), gather x, stage products ----------
        // Indices are clamped instead of predicated so the loads of one round
        // issue back to back (no per-element branch + wait).
        if (cnt > 0) {
            const int last = cnt - 1;
            for (int k = tid; k < cnt; k += 4 * SPMV_BLOCK) {
                int kk[4];
                double a[4];
                int c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = k + j * SPMV_BLOCK;
                    kk[j] = t < last ? t : last;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = val[p0 + kk[j]];
                    c[j] = col[p0 + kk[j]];
                }
                double xv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = x[c[j]];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = k + j * SPMV_BLOCK;
                    if (t < cnt) prod[t] = a[j] * (xv[j] * sx);
                }
            }
        }
        __syncthreads();

        // ---- phase 2: reduce the short rows out of LDS ---------------------
        const int nr = r1s - r0;
        if (nr > 0) {
            // lanes per row from the block's mean row length: <= 16 nonzeros -> one lane,
            // plain left-to-right sum (bit-identical to the reference's COO-order row
            // sums); longer rows get 2..64 lanes and a shuffle tree.
            const int avg = cnt / nr;
            int G = 1;
            while (G < WAVE && avg > 16 * G) G <<= 1;  // uniform
            const int gl = tid & (G - 1), gid = tid / G, ngroups = SPMV_BLOCK / G;
            for (int r = r0 + gid; r < r1s; r += ngroups) {
                const int s0 = (int)(rowptr[r] - p0), s1 = (int)(rowptr[r + 1] - p0);
                double s = 0.0;
                for (int k = s0 + gl; k < s1; k += G) s = s + prod[k];
                for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
                if (gl == 0) {
                    const double yn = cy * (y[r] * sy) + s;
                    y[r] = yn;
                    sq += yn * yn;
                }
            }
        }

        // ---- phase 3: a long last row, split across the workgroup ----------
        if (has_long) {
            const OffT len = pend - plast;
            const OffT lastk = len - 1;
            double s = 0.0;
            for (OffT k = tid; k < len; k += 4 * SPMV_BLOCK) {
                OffT kk[4];
                double a[4];
                int c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const OffT t = k + j * SPMV_BLOCK;
                    kk[j] = t < lastk ? t : lastk;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = val[plast + kk[j]];
                    c[j] = col[plast + kk[j]];
                }
                double xv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = x[c[j]];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const OffT t = k + j * SPMV_BLOCK;
                    if (t < len) s = s + a[j] * (xv[j] * sx);
                }
            }
            const double tot = block_sum<SPMV_BLOCK>(s, red);
            if (tid == 0) {
                const int r = r1 - 1;
                const double yn = cy * (y[r] * sy) + tot;
                y[r] = yn;
                sq += yn * yn;
            }
        }
        __syncthreads();  // prod[] is rewritten by the next row block
    }

    const double tot = block_sum<SPMV_BLOCK>(sq, red);
    if (tid == 0) partials[blockIdx.x] = tot;
}

}  // namespace lsqrhip
