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
// partial of sum(y^2), reduced in fixed order by its consumers => deterministic.
//
// Coefficients come in one of two ways:
//   explicit  `coef` points at an SpmvCoef (aprod, initialisation, sharded stages);
//   lazy      `pin` points at the partials of the PREVIOUS SpMV of the iteration: every
//             workgroup reduces them itself (fixed order, so all workgroups and the scalar
//             kernels get identical bits), takes nrm = sqrt(sum) and uses
//                 sx = 1/nrm (1 if nrm = 0),  cy = -nrm,  sy = slot_in->scale,
//             i.e. mode 1 derives alpha from the mode-2 partials and mode 2 derives beta from
//             the mode-1 partials, with no scalar kernel on the critical path between them.
//             Workgroup 0 publishes (nrm, 1/nrm) in slot_out for the next kernel's sy.
//
// Measured at config 2 (88 MB per launch) with this kernel: ~17 us, 16 us with 16-bit columns,
// 14.4 us with the value dictionary (15.7 before the first round skipped its all-clamped slots) -- and ~4.2 us per resident round of windows whatever the
// bytes: a trip through a window is a chain of dependent phases.  Matrices with
// short even rows therefore go to the sliced-ELL kernel (sell.h: 10 us), dense scattered rows to
// the LDS-panel kernel (xl.h); this one serves every other shape.  Sweeps: window 512 / 1024 /
// 2048 / 4096 -> 23 / 17 / 21 / 31 us; one workgroup per row block instead of a persistent
// grid: +-2 % on banded, +13 % on power-law rows.
//
// Algorithmic HBM bytes per launch:  12*nnz + P*(rows+1) + 8*cols + 16*rows.
#pragma once

#include <type_traits>

#include "common.h"
#include "scalar.h"
#include "state.h"
#include "valdict.h"
#include "vec.h"

namespace lsqrhip {

constexpr int SPMV_BLOCK = 256;
constexpr int SPMV_C = 1024;          // window size in work units (nonzeros + rows)
constexpr int SPMV_LDS = 2 * SPMV_C;  // products staged per row block (doubles)
constexpr int SPMV_MAX_GRID = 2048;   // 8 workgroups per CU x 256 CUs
// Nonzeros per lane before a row gets more lanes in phase 2.  16 keeps rows of <= 16 nonzeros on ONE
// lane (the reference's left-to-right sum, bit for bit); panelled products regroup a row's sum by
// panel anyway and take more lanes earlier (8: 4M x 1M x 100 -2 %, power law -2 %; 4: power law
// mode 1 -7 % but 4M x 1M mode 2 +4 %).
constexpr int SPMV_G_PANEL = 8;
static_assert(SPMV_BLOCK == VEC_BLOCK, "the fused update runs k_update's blocks");

// rb[k] = first row r in [0, m] with rowptr[r] + r >= k*C ; rb[nblk] = m.  C = window size in work
// units (SPMV_C, or XLW_C for the wave-window layout of xl.h).
template <typename OffT>
__global__ void k_row_blocks(const OffT *__restrict__ rowptr, int m, int64_t nblk, int *__restrict__ rb, int C)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nblk) return;
    if (k == nblk) {
        rb[k] = m;
        return;
    }
    const int64_t target = k * (int64_t)C;
    int lo = 0, hi = m;  // answer in [lo, hi]
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if ((int64_t)rowptr[mid] + mid >= target) hi = mid;
        else lo = mid + 1;
    }
    rb[k] = lo;
}

// One descriptor per row block, built once (k_block_desc): everything the kernel needs to
// start streaming, fetched with ONE scalar load instead of the dependent chain
// rb[b] -> rowptr[r0], rowptr[r1-1], rowptr[r1].
struct alignas(32) RowBlock {
    long long p0;     // first nonzero of the block
    long long plast;  // first nonzero of the last row
    long long pend;   // one past the last nonzero
    int r0, r1;       // rows [r0, r1)
};

// Column span of every row block: cbase[b] = smallest column; *too_wide |= 1 if a block spans
// 65536 columns or more.  One workgroup per row block.
__global__ __launch_bounds__(256) void k_block_colspan(const RowBlock *__restrict__ blk, int64_t nblk,
                                                       const int *__restrict__ col, int *__restrict__ cbase,
                                                       int *__restrict__ too_wide)
{
    __shared__ int smin[256], smax[256];
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const RowBlock d = blk[b];
        int lo = 0x7fffffff, hi = -1;
        if (d.r0 < d.r1)
            for (long long k = d.p0 + threadIdx.x; k < d.pend; k += 256) {
                const int c = col[k];
                lo = c < lo ? c : lo;
                hi = c > hi ? c : hi;
            }
        smin[threadIdx.x] = lo;
        smax[threadIdx.x] = hi;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) {
                smin[threadIdx.x] = min(smin[threadIdx.x], smin[threadIdx.x + off]);
                smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + off]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const int mn = smax[0] >= 0 ? smin[0] : 0;
            cbase[b] = mn;
            if (smax[0] >= 0 && smax[0] - mn >= 65536) atomicOr(too_wide, 1);
        }
        __syncthreads();
    }
}

// col16[k] = col[k] - cbase[block of k]
__global__ __launch_bounds__(256) void k_col_to16(const RowBlock *__restrict__ blk, int64_t nblk,
                                                  const int *__restrict__ col, const int *__restrict__ cbase,
                                                  unsigned short *__restrict__ col16)
{
    for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        const RowBlock d = blk[b];
        if (d.r0 >= d.r1) continue;
        const int cb = cbase[b];
        for (long long k = d.p0 + threadIdx.x; k < d.pend; k += 256) col16[k] = (unsigned short)(col[k] - cb);
    }
}

template <typename OffT>
__global__ void k_block_desc(const OffT *__restrict__ rowptr, const int *__restrict__ rb, int64_t nblk,
                             RowBlock *__restrict__ blk)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    RowBlock d;
    d.r0 = rb[b];
    d.r1 = rb[b + 1];
    if (d.r0 < d.r1) {
        d.p0 = (long long)rowptr[d.r0];
        d.plast = (long long)rowptr[d.r1 - 1];
        d.pend = (long long)rowptr[d.r1];
    } else {
        d.p0 = d.plast = d.pend = 0;
    }
    blk[b] = d;
}

// Column panels (PANEL = true).  When x does not fit an XCD's 4 MB L2 and the columns are not
// local (random / power-law matrices), every 8-byte gather of x pulls a whole sector from
// beyond L2: measured 1.2 / 0.88 / 0.65 TB/s of algorithmic bytes at n = 1e6 / 2e6 / 1e7.
// The build then stores the CSR of the PANEL-STACKED matrix [A_0; A_1; ...; A_{P-1}], A_p =
// the columns of panel p (a slice of x of ~2.5 MB): virtual row v = p * rows + r.  The same
// kernel sweeps the virtual rows; because the work list is panel-major and each XCD owns one
// contiguous eighth of it (common.h), an XCD gathers from one x slice at a time and that
// slice lives in its L2.  The kernel writes the per-panel row sums z[v]; k_panel_combine
// then forms y_r <- cy (y_r sy) + sum_p z[p*rows + r] in fixed panel order (deterministic)
// and produces the partials of sum(y^2).
//
// launch bound 8 waves/SIMD: the kernel must stay within 64 VGPRs (66 cost a whole
// workgroup per CU: 18.9 us instead of 17.2 at config 2)
//
// C16 = true: 16-bit column indices relative to the row block's smallest column (cbase[b]).
// Chosen at build time when EVERY row block spans fewer than 65536 columns (banded / local
// matrices): 2 instead of 4 bytes per nonzero on a kernel that is bound by bytes.
//
// V8 = true: one-byte value codes into the matrix-wide dictionary dict[256] (valdict.h), held in
// LDS: 1 instead of 8 bytes per nonzero when the matrix has <= 256 distinct values.
// UPD = true: the launch also carries the x/w update of the previous iteration (vec.h UpdArgs).
//
// XL = true (PANEL only): LDS-resident panels.  When the panel width is <= XL_COLS the current
// panel's slice of x (already multiplied by sx) is held in LDS and the gathers never leave the
// CU: the product is then bound by streaming (val, col), not by the ~0.29 L1 misses per clock a
// CU can retire.  Needs >= ~1.3 nonzeros per virtual row to pay for the partial
// sums of its many narrow panels: BASELINE config 3 at its literal 1000 per row.  72 KB of LDS
// per workgroup -> 2 workgroups per CU.  The few entries of a window that reaches into the
// next panel are gathered from global memory.
constexpr int XL_COLS = 7168;  // 56 KB of x per panel

struct XlArgs {
    int rows;   // real rows of the product (virtual row v = panel * rows + r)
    int pw;     // panel width in columns (<= XL_COLS when XL)
    int ncols;  // columns of the matrix
    const unsigned char *skew;  // PANEL: skew[b] != 0 = window b holds a segment > SPMV_LONGCUT (or null)
    const unsigned short *rel16;  // xl.h C16: row starts relative to their window's first nonzero
};

// Skewed windows (power-law rows cut by panels: a third of the nonzeros sit in segments of more than
// 64 while the window's mean is 3).  Phase 2 gives every row the lanes the window's MEAN calls
// for, so one lane would walk such a segment alone while the workgroup waits.  In windows the
// build has flagged, segments longer than SPMV_LONGCUT are set aside in phase 2 and summed
// afterwards by a whole wave each (stride-64 partial sums + shuffle tree).  Panelled products only:
// their kernel carries no sum of squares, so which wave takes which segment cannot change a bit.
constexpr int SPMV_LONGCUT = 128;
constexpr int SPMV_MAXLONG = 2 * SPMV_C / SPMV_LONGCUT;  // segments > LONGCUT in one window (< 2C nonzeros)

template <typename OffT>
__global__ void k_block_skew(const OffT *__restrict__ rowptr, const RowBlock *__restrict__ blk, int64_t nblk,
                             unsigned char *__restrict__ skew, int *__restrict__ any)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const RowBlock d = blk[b];
    int r1s = d.r1;
    if (d.r0 < d.r1 && d.pend - d.plast >= (long long)SPMV_C) r1s = d.r1 - 1;  // phase 3 takes that one
    unsigned char f = 0;
    for (int r = d.r0; r < r1s; ++r)
        if ((long long)rowptr[r + 1] - (long long)rowptr[r] > SPMV_LONGCUT) {
            f = 1;
            break;
        }
    skew[b] = f;
    if (f) *any = 1;
}

// VT = storage type of x, y and the values (double; float for a REAL32 handle -- never with PANEL, whose
// y is the double array Z).  Products and sums are formed in binary64 either way.
template <typename OffT, bool PANEL, bool C16, bool V8, bool UPD, bool XL = false, typename VT = double>
__global__ __launch_bounds__(SPMV_BLOCK, XL ? 2 : 8) void k_spmv_fused(
    const OffT *__restrict__ rowptr, const void *__restrict__ colv, const int *__restrict__ cbase,
    const void *__restrict__ valv, const double *__restrict__ dict, const RowBlock *__restrict__ blk, int64_t nblk, const VT *__restrict__ x, VT *__restrict__ y,
    const SpmvCoef *__restrict__ coef, const int *__restrict__ stop, double *__restrict__ partials,
    const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in,
    NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, UpdArgs upd, XlArgs xa, NScale nsc)
{
    __shared__ double prod[SPMV_LDS];
    __shared__ double xs[XL ? XL_COLS : 1];
    __shared__ double red[SPMV_BLOCK / WAVE + 1];
    __shared__ double sdict[V8 ? VD_MAX : 1];
    __shared__ int longlist[PANEL ? SPMV_MAXLONG : 1];
    __shared__ int nlong;
    // One extra workgroup carries scalar work (scalar.h "riders").  It is block 0, the first
    // one dispatched, so it runs beside the SpMV from the start (as the LAST block of a grid
    // that exceeds the resident slots it would only start when the first SpMV block retires).
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;     // SpMV workgroups
    const int wg = (int)blockIdx.x - shift;     // this workgroup's index among them
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    if (*stop != 0) return;
    const int tid = threadIdx.x;
    const int *__restrict__ col = static_cast<const int *>(colv);
    const unsigned short *__restrict__ col16 = static_cast<const unsigned short *>(colv);
    const VT *__restrict__ val = static_cast<const VT *>(valv);
    const unsigned char *__restrict__ val8 = static_cast<const unsigned char *>(valv);
    if (V8) sdict[tid] = dict[tid];  // visible after the first barrier below

    double sx, sy, cy;
    if (pin != nullptr) {  // lazy coefficients (uniform branch)
        const double nrm = sqrt(block_sum_all<SPMV_BLOCK>(pin, npin, red)) * nsc.inv;
        if (skip_if_zero && !(nrm > 0.0)) {  // mode 2 is skipped when beta == 0 (:691)
            if (wg == 0 && tid == 0) {
                slot_out->nrm = nrm;
                slot_out->scale = 1.0;
            }
            return;
        }  // mode 2 is skipped when beta == 0 (:691)
        sx = nrm > 0.0 ? 1.0 / nrm : 1.0;
        cy = -nrm;
        sy = slot_in->scale;
        if (wg == 0 && tid == 0) {
            slot_out->nrm = nrm;
            slot_out->scale = sx;
        }
        if (UPD && upd.on == 2) {  // the first launch of a solve: w <- v / alpha  (src/lsqr.f90:641-644)
            const XcdRange ur = xcd_range(upd.ugrid, nwg, wg);
            for (int ub = (int)ur.first; ub < (int)ur.end; ub += (int)ur.stride)
                winit_block<VT>((VT *)upd.w, (const VT *)upd.V, upd.n, sx, ub, upd.ugrid);
        } else if (UPD && upd.on) {  // x/w update of the previous iteration (vec.h UpdArgs)
            const double beta = slot_in->nrm;
            double alpha = nrm, sv = sx;
            if (!(beta > 0.0)) {  // mode 2 was skipped (src/lsqr.f90:691): alpha, v unchanged
                alpha = upd.alpha_prev->nrm;
                sv = upd.alpha_prev->scale;
            }
            const LsqrState *ust = upd.st;
            const Rot rt = rot_step(ust->rhobar2[upd.par], ust->phibar2[upd.par], ust->damp, ust->damped, alpha, beta);
            const bool wantse = ust->wantse != 0;
            // XCD-contiguous blocks, like the rows below: the slice of V this XCD updates from is
            // the slice its rows gather from (one trip from beyond L2 instead of two)
            const XcdRange ur = xcd_range(upd.ugrid, nwg, wg);
            for (int ub = (int)ur.first; ub < (int)ur.end; ub += (int)ur.stride) {
                const double tot = update_block<VT>((VT *)upd.x, (VT *)upd.w, (const VT *)upd.V, (VT *)upd.se, upd.n, rt.t1,
                                                    rt.t2, rt.t3, sv, wantse, ub, upd.ugrid, red);
                if (tid == 0) upd.pout[ub] = tot;
            }
        }
    } else {
        if (coef->skip != 0) return;
        sx = coef->sx;
        sy = coef->sy;
        cy = coef->cy;
    }
    if (V8) __syncthreads();

    double sq = 0.0;  // this thread's share of sum(y_new^2)
    int xpid = -1, xbase = 0;  // XL: panel whose slice is in xs[], and its first column
    // x_j * sx for column j: from the LDS slice when the column is inside it
    auto gx = [&](int cj) -> double {
        if (XL) {
            const unsigned off = (unsigned)(cj - xbase);
            if (off < (unsigned)xa.pw) return xs[off];
        }
        return (double)x[cj] * sx;
    };

    // Independent loads are issued up front: the descriptor of the NEXT block, this block's
    // (val, col), and the row pointers + y of the row this lane will reduce.
    const XcdRange xr = xcd_range(nblk, nwg, wg);
    int64_t b = xr.first;
    RowBlock d;
    int cbn = 0;
    if (b < xr.end) {
        d = blk[b];
        if (C16) cbn = cbase[b];
    }
    for (; b < xr.end; b += xr.stride) {
        const RowBlock cur = d;
        const int cb = cbn;
        if (b + xr.stride < xr.end) {  // prefetch (uniform index)
            d = blk[b + xr.stride];
            if (C16) cbn = cbase[b + xr.stride];
        }
        const int r0 = cur.r0, r1 = cur.r1;
        if (r0 >= r1) continue;  // uniform
        const bool skewed = PANEL && xa.skew != nullptr && xa.skew[b] != 0;  // uniform
        if (skewed && tid == 0) nlong = 0;  // visible after the barrier that ends phase 1
        if (XL) {  // this window's panel slice of x (times sx) into LDS, when it changes
            const int pid = r0 / xa.rows;
            if (pid != xpid) {  // uniform; the barrier at the end of the last trip covers xs
                xpid = pid;
                xbase = pid * xa.pw;
                for (int i = tid; i < xa.pw; i += SPMV_BLOCK) {
                    const int cx = xbase + i;
                    xs[i] = cx < xa.ncols ? (double)x[cx] * sx : 0.0;
                }
                __syncthreads();
            }
        }
        const OffT p0 = (OffT)cur.p0, plast = (OffT)cur.plast, pend = (OffT)cur.pend;
        const bool has_long = (pend - plast) >= (OffT)SPMV_C;
        const int r1s = has_long ? r1 - 1 : r1;
        const int cnt = (int)((has_long ? plast : pend) - p0);  // < 2C by construction
        const int nr = r1s - r0;

        // lanes per row from the block's mean row length: <= 16 nonzeros -> one lane, plain
        // left-to-right sum (bit-identical to the reference's COO-order row sums); longer
        // rows get 2..64 lanes and a shuffle tree.
        int G = 1;
        if (nr > 0) {
            const int avg = cnt / nr;
            while (G < WAVE && avg > (PANEL ? SPMV_G_PANEL : 16) * G) G <<= 1;  // uniform
        }
        const int gl = tid & (G - 1), gid = tid / G, ngroups = SPMV_BLOCK / G;

        // ---- phase 1 loads: (val, col) of the block; indices clamped, not predicated, so
        // the loads of a round issue back to back (no per-element branch + wait).  The first
        // round runs as straight-line code for exactly the NS = ceil(cnt / 256) slots that hold
        // nonzeros (uniform switch): with 2-3 nonzeros per (row, panel) a window holds ~730
        // nonzeros, and a fourth, all-clamped slot is a quarter more memory instructions for the
        // texture addresser, which is what bounds this kernel (PMC: TA busy 76 % of the launch).
        const int last = cnt > 0 ? cnt - 1 : 0;
        // ---- early loads for phase 2: this lane's first row ---------------------------
        const int rfirst = r0 + gid;
        const bool have_row = rfirst < r1s;
        const int rclamp = have_row ? rfirst : r0;
        OffT q0 = 0, q1 = 0, q0b = 0, q1b = 0;
        double y0 = 0.0, y0b = 0.0;
        auto stage = [&](auto ns_tag) {
            constexpr int NS = decltype(ns_tag)::value;
            constexpr int NA = NS > 0 ? NS : 1;
            int kk[NA];
            double a[NA];
            int c[NA];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int t = tid + j * SPMV_BLOCK;
                kk[j] = t < last ? t : last;
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                a[j] = V8 ? sdict[val8[p0 + kk[j]]] : (double)val[p0 + kk[j]];
                c[j] = C16 ? cb + (int)col16[p0 + kk[j]] : col[p0 + kk[j]];
            }
            if (nr > 0) {
                q0 = rowptr[rclamp];
                q1 = rowptr[rclamp + 1];
                if (!PANEL) y0 = (double)y[rclamp];
                if (nr > ngroups) {  // uniform: this lane's SECOND round too (clamped)
                    const int r2 = rfirst + ngroups < r1s ? rfirst + ngroups : rclamp;
                    q0b = rowptr[r2];
                    q1b = rowptr[r2 + 1];
                    if (!PANEL) y0b = (double)y[r2];
                }
            }
            double xv[NA];
#pragma unroll
            for (int j = 0; j < NS; ++j) xv[j] = gx(c[j]);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int t = tid + j * SPMV_BLOCK;
                if (t < cnt) prod[t] = a[j] * xv[j];
            }
        };
        if (cnt <= 0) stage(std::integral_constant<int, 0>{});
        else if (cnt <= SPMV_BLOCK) stage(std::integral_constant<int, 1>{});
        else if (cnt <= 2 * SPMV_BLOCK) stage(std::integral_constant<int, 2>{});
        else if (cnt <= 3 * SPMV_BLOCK) stage(std::integral_constant<int, 3>{});
        else stage(std::integral_constant<int, 4>{});
        if (cnt > 4 * SPMV_BLOCK) {  // cnt in (1024, 2C): the rest in clamped rounds of four
            int kk[4];
            double a[4], xv[4];
            int c[4];
            for (int k = tid + 4 * SPMV_BLOCK; k < cnt; k += 4 * SPMV_BLOCK) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = k + j * SPMV_BLOCK;
                    kk[j] = t < last ? t : last;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] = V8 ? sdict[val8[p0 + kk[j]]] : (double)val[p0 + kk[j]];
                    c[j] = C16 ? cb + (int)col16[p0 + kk[j]] : col[p0 + kk[j]];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = gx(c[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = k + j * SPMV_BLOCK;
                    if (t < cnt) prod[t] = a[j] * xv[j];
                }
            }
        }
        __syncthreads();

        // ---- phase 2: reduce the short rows out of LDS ---------------------------------
        if (have_row) {
            int r = rfirst;
            for (;;) {
                const int s0 = (int)(q0 - p0), s1 = (int)(q1 - p0);
                double s = 0.0;
                const bool aside = skewed && s1 - s0 > SPMV_LONGCUT;  // the same for the G lanes of a row
                if (!aside) {
                    for (int k = s0 + gl; k < s1; k += G) s = s + prod[k];
                    for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
                }
                if (gl == 0) {
                    if (aside) {
                        longlist[atomicAdd(&nlong, 1)] = r;
                    } else if (PANEL) {
                        y[r] = (VT)s;  // z[v]: raw sum of this (panel, row) segment
                    } else {
                        const VT yn = (VT)(cy * (y0 * sy) + s);   // as stored (rounded to float for a REAL32 handle)
                        y[r] = yn;
                        const double ys = (double)yn * nsc.s;
                        sq += ys * ys;
                    }
                }
                r += ngroups;
                if (r >= r1s) break;
                if (r == rfirst + ngroups) {  // second round: already here
                    q0 = q0b;
                    q1 = q1b;
                    y0 = y0b;
                } else {
                    q0 = rowptr[r];
                    q1 = rowptr[r + 1];
                    if (!PANEL) y0 = (double)y[r];
                }
            }
        }

        // ---- phase 2b: the segments set aside in a skewed window, one wave each --------
        if (skewed) {
            __syncthreads();
            const int nl = nlong;
            const int lane = tid & (WAVE - 1);
            for (int i = tid >> 6; i < nl; i += SPMV_BLOCK / WAVE) {
                const int r = longlist[i];
                const int s0 = (int)(rowptr[r] - p0), s1 = (int)(rowptr[r + 1] - p0);
                double s = 0.0;
                for (int k = s0 + lane; k < s1; k += WAVE) s = s + prod[k];
                s = wave_sum(s);
                if (lane == 0) y[r] = (VT)s;
            }
        }

        // ---- phase 3: a long last row, split across the workgroup ----------------------
        if (has_long) {
            const OffT len = pend - plast;
            const OffT lastk = len - 1;
            double s = 0.0;
            for (OffT k = tid; k < len; k += 4 * SPMV_BLOCK) {
                OffT kl[4];
                double al[4];
                int cl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const OffT t = k + j * SPMV_BLOCK;
                    kl[j] = t < lastk ? t : lastk;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    al[j] = V8 ? sdict[val8[plast + kl[j]]] : (double)val[plast + kl[j]];
                    cl[j] = C16 ? cb + (int)col16[plast + kl[j]] : col[plast + kl[j]];
                }
                double xl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xl[j] = gx(cl[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const OffT t = k + j * SPMV_BLOCK;
                    if (t < len) s = s + al[j] * xl[j];
                }
            }
            const double tot = block_sum<SPMV_BLOCK>(s, red);
            if (tid == 0) {
                const int r = r1 - 1;
                if (PANEL) {
                    y[r] = (VT)tot;
                } else {
                    const VT yn = (VT)(cy * ((double)y[r] * sy) + tot);
                    y[r] = yn;
                    const double ys = (double)yn * nsc.s;
                    sq += ys * ys;
                }
            }
        }
        __syncthreads();  // prod[] is rewritten by the next row block
    }

    if (!PANEL) {
        const double tot = block_sum<SPMV_BLOCK>(sq, red);
        if (tid == 0) partials[wg] = tot;
    }
}

// y_r <- cy (y_r sy) + sum_{p < P} z[p*rows + r]   (fixed panel order); partials of sum(y^2).
// Coefficients as in k_spmv_fused: explicit (coef) or lazy (pin/slot_in); only cy, sy are used.
__global__ __launch_bounds__(SPMV_BLOCK) void k_panel_combine(
    double *__restrict__ y, const double *__restrict__ z, int rows, int P,
    const SpmvCoef *__restrict__ coef, const int *__restrict__ stop, double *__restrict__ partials,
    const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in, int skip_if_zero, NScale nsc)
{
    if (*stop != 0) return;
    __shared__ double red[SPMV_BLOCK / WAVE + 1];
    double sy, cy;
    if (pin != nullptr) {
        const double nrm = sqrt(block_sum_all<SPMV_BLOCK>(pin, npin, red)) * nsc.inv;
        if (skip_if_zero && !(nrm > 0.0)) return;
        cy = -nrm;
        sy = slot_in->scale;
    } else {
        if (coef->skip != 0) return;
        sy = coef->sy;
        cy = coef->cy;
    }
    double sq = 0.0;
    const int64_t stride = (int64_t)gridDim.x * SPMV_BLOCK;
    for (int64_t r = (int64_t)blockIdx.x * SPMV_BLOCK + threadIdx.x; r < rows; r += stride) {
        const double yold = y[r];
        double s = 0.0;
        int p = 0;
        for (; p + 4 <= P; p += 4) {  // 4 independent loads in flight, added in panel order
            const double v0 = z[(int64_t)p * rows + r], v1 = z[(int64_t)(p + 1) * rows + r];
            const double v2 = z[(int64_t)(p + 2) * rows + r], v3 = z[(int64_t)(p + 3) * rows + r];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; p < P; ++p) s = s + z[(int64_t)p * rows + r];
        const double yn = cy * (yold * sy) + s;
        y[r] = yn;
        const double ys = yn * nsc.s;
        sq += ys * ys;
    }
    const double tot = block_sum<SPMV_BLOCK>(sq, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// The tail of a batch in the fused schedule: the x/w update of the batch's last iteration with the rotation
// taken lazily, exactly as the next mode-1 launch would take it (sell.h sell_prologue / spmv.h: alpha from the
// mode-2 partials, beta from the slot the mode-2 launch published, rhobar / phibar by parity), and steps 1+2 of
// that iteration as the rider (block 0) -- one launch where k_s12 and then k_update ran.  Same functions on the
// same inputs: the same bits.
// `copy_out`: the batch ends with the snapshot the host polls, and x may be wanted at an address of the caller's
// (LsqrState.xout, device-resident solves).  The new x is then stored there as well as in X; and if the stop flag is
// already up -- the solve ended earlier in this batch, or in the batch before this look-ahead one -- X is final and is
// copied there.  Either way x is at xout when the batch ends, whether or not step 3 of this iteration then stops the
// solve: vec.h k_out_copy (a launch of its own, 4.8 us of a 20-iteration solve at config 2) is not needed behind it.
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_update_lazy(const double *__restrict__ pin, int npin,
                                                           const NormSlot *__restrict__ slot_in, UpdArgs upd, Rider rider,
                                                           NScale nsc, const int *__restrict__ stop, int copy_out)
{
    __shared__ double red[VEC_BLOCK / WAVE + 1];
    const int wg = (int)blockIdx.x - 1;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    VT *xout = copy_out ? static_cast<VT *>(upd.st->xout) : nullptr;
    if (*stop != 0) {
        if (xout != nullptr) {
            const VT *X = static_cast<const VT *>(upd.x);
            const int64_t stride = (int64_t)upd.ugrid * VEC_BLOCK;
            for (int64_t i = (int64_t)wg * VEC_BLOCK + threadIdx.x; i < upd.n; i += stride) xout[i] = X[i];
        }
        return;
    }
    const double nrm = sqrt(block_sum_all<VEC_BLOCK>(pin, npin, red)) * nsc.inv;
    const double beta = slot_in->nrm;
    double alpha = nrm, sv = nrm > 0.0 ? 1.0 / nrm : 1.0;
    if (!(beta > 0.0)) {  // mode 2 was skipped (src/lsqr.f90:691): alpha, v unchanged
        alpha = upd.alpha_prev->nrm;
        sv = upd.alpha_prev->scale;
    }
    const LsqrState *ust = upd.st;
    const Rot rt = rot_step(ust->rhobar2[upd.par], ust->phibar2[upd.par], ust->damp, ust->damped, alpha, beta);
    const double tot = update_block<VT>((VT *)upd.x, (VT *)upd.w, (const VT *)upd.V, (VT *)upd.se, upd.n, rt.t1, rt.t2,
                                        rt.t3, sv, ust->wantse != 0, wg, upd.ugrid, red, xout);
    if (threadIdx.x == 0) upd.pout[wg] = tot;
}

}  // namespace lsqrhip
