// sell.h -- sliced-ELL SpMV for matrices with short, even rows (aprod mode 1 / mode 2).
//
// Same contract as spmv.h's k_spmv_fused (reference src/lsqr.f90:166-174 / :186-194 fused with
// the dscal before and the dnrm2 after, :681-683 / :692-695):
//
//     y_i  <-  cy * (y_i * sy)  +  sum_j A_ij * (x_j * sx)        partial += y_i^2
//
// Why a second layout.  The row-window kernel (spmv.h) handles any degree distribution, but a
// trip through one window is a chain of dependent steps (descriptor -> (val, col) -> x gather
// -> LDS -> barrier -> row sums -> barrier): ~4 us per resident round of windows whatever the
// bytes (profiles/r01/sweep_prefetch_negative.txt).  When every row is short and the rows of a
// 64-row slice have about the same length -- stencils, meshes, BASELINE.json configs[1] -- a
// slice stored COLUMN-major needs none of that: lane i of a wave owns row i of the slice, the
// k-th load of the wave reads the k-th nonzero of 64 consecutive rows (one contiguous
// segment), each lane adds its own products left to right in registers.  No LDS staging, no
// barriers, no row pointers; all W loads of a slice are issued before the first use.
//
// Layout (built by k_sell_* below from the CSR of csr_build.h):
//   slice s = rows [64 s, 64 s + 64);  W_s = longest row of the slice;
//   soff[s] = first element of slice s (in elements, 64 W_s per slice), soff[nslices] = total;
//   element (row i of slice, k)  at  soff[s] + 64 k + i;   rlen[r] = length of row r (<= 64);
//   padding slots (k >= rlen) hold the slice's smallest column and value 0 -- they are loaded
//   but never added (the add is predicated on k < rlen, so no 0*inf, no -0 + 0 differences).
// Every row sum is the plain left-to-right sum in COO order starting from 0, i.e. bit-identical
// to the reference's and to the window kernel's one-lane-per-row path.
//
// Chosen at build time when mean row length <= 24, no row is longer than 64, padding costs
// <= 12.5 % and the columns are local (every slice spans < 65536 columns; with scattered
// columns the x gathers dominate and the layout buys nothing).  LSQRHIP_SELL=0 disables,
// =1 drops the locality requirement.  Columns are 16-bit relative to the slice's
// smallest column when every slice spans < 65536 columns; values are one-byte dictionary
// codes when the matrix has a dictionary (valdict.h).
#pragma once

#include "common.h"
#include "scalar.h"
#include "state.h"
#include "valdict.h"
#include "vec.h"

namespace lsqrhip {

constexpr int SELL_BLOCK = 256;                 // 4 slices per workgroup trip
constexpr int SELL_SLICES = SELL_BLOCK / WAVE;
constexpr int SELL_MAX_W = 64;
constexpr int SELL_MAX_GRID = 1536;             // 6 workgroups per CU x 256 CUs (lsqrhip.hip)
constexpr int SELL_SHARE_K = 6;                 // partial sums per thread requested at the top (<= 1536 of them)
constexpr int SELLP_K = 5;                      // nonzeros per 16-byte record of the packed layout

// width64[s] = 64 * W_s (elements of slice s); stats[0] += 64 W_s, stats[1] = max W,
// stats[3] += 64 * ceil(W_s / 5) (records of the packed layout below).
__global__ __launch_bounds__(256) void k_sell_width(const int *__restrict__ rowptr, int rows, int nslices,
                                                    unsigned *__restrict__ width64,
                                                    unsigned long long *__restrict__ stats)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)(r >> 6);
    int len = 0;
    if (r < rows) len = rowptr[r + 1] - rowptr[r];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off, WAVE));
    if ((threadIdx.x & (WAVE - 1)) == 0 && s < nslices) {
        width64[s] = (unsigned)len * 64u;
        atomicAdd(&stats[0], (unsigned long long)len * 64ull);
        atomicMax(&stats[1], (unsigned long long)len);
        atomicAdd(&stats[3], (unsigned long long)((len + SELLP_K - 1) / SELLP_K) * 64ull);
    }
}

// width64[s] (64 W_s) -> 64 * ceil(W_s / 5): records of slice s in the packed layout
__global__ __launch_bounds__(256) void k_sellp_chunks(unsigned *__restrict__ width64, int nslices)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslices) width64[s] = (((width64[s] >> 6) + SELLP_K - 1) / SELLP_K) * 64u;
}

// cbaseS[s] = smallest column of slice s (0 if empty); stats[2] |= 1 if a slice spans >= 65536 columns.
__global__ __launch_bounds__(256) void k_sell_colspan(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      int rows, int nslices, int *__restrict__ cbaseS,
                                                      unsigned long long *__restrict__ stats)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)(r >> 6);
    int lo = 0x7fffffff, hi = -1;
    if (r < rows) {
        const int q0 = rowptr[r], q1 = rowptr[r + 1];
        for (int k = q0; k < q1; ++k) {
            const int c = col[k];
            lo = min(lo, c);
            hi = max(hi, c);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, WAVE));
        hi = max(hi, __shfl_xor(hi, off, WAVE));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && s < nslices) {
        cbaseS[s] = hi >= 0 ? lo : 0;
        if (hi >= 0 && hi - lo >= 65536) atomicOr(&stats[2], 1ull);
    }
}

// CSR rows -> column-major slices.  One thread per (padded) row.
template <bool C16, bool V8>
__global__ __launch_bounds__(256) void k_sell_fill(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                   const double *__restrict__ val,
                                                   const unsigned *__restrict__ soff, const int *__restrict__ cbaseS,
                                                   const unsigned long long *__restrict__ dict_bits, int nd, int rows,
                                                   int nslices, void *__restrict__ scolv, void *__restrict__ svalv,
                                                   unsigned char *__restrict__ rlen)
{
    __shared__ unsigned long long tab[VD_MAX];
    if (V8) {
        if ((int)threadIdx.x < nd) tab[threadIdx.x] = dict_bits[threadIdx.x];
        __syncthreads();
    }
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)(r >> 6), lane = (int)(r & 63);
    if (s >= nslices) return;
    const unsigned o0 = soff[s];
    const int W = (int)((soff[s + 1] - o0) >> 6);
    const int cb = cbaseS[s];
    int q0 = 0, len = 0;
    if (r < rows) {
        q0 = rowptr[r];
        len = rowptr[r + 1] - q0;
        rlen[r] = (unsigned char)len;
    }
    unsigned short *sc16 = static_cast<unsigned short *>(scolv);
    int *sc32 = static_cast<int *>(scolv);
    unsigned char *sv8 = static_cast<unsigned char *>(svalv);
    double *sv = static_cast<double *>(svalv);
    for (int k = 0; k < W; ++k) {
        const size_t idx = (size_t)o0 + (size_t)k * 64 + lane;
        int c = cb;
        double v = 0.0;
        if (k < len) {
            c = col[q0 + k];
            v = val[q0 + k];
        }
        if (C16) sc16[idx] = (unsigned short)(c - cb);
        else sc32[idx] = c;
        if (V8) {
            unsigned char code = 0;
            if (k < len) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
                int lo = 0, hi = nd - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (tab[mid] < bits) lo = mid + 1;
                    else hi = mid;
                }
                code = (unsigned char)lo;
            }
            sv8[idx] = code;
        } else {
            sv[idx] = v;
        }
    }
}

// U columns of one slice, starting at element `e` (= soff + 64 k0 + lane): all loads first,
// then the predicated left-to-right adds.
template <int U, bool C16, bool V8, typename VT, bool NT>
__device__ __forceinline__ double sell_chunk(double sum, size_t e, int k0, int len, int cb,
                                             const int *__restrict__ sc32, const unsigned short *__restrict__ sc16,
                                             const VT *__restrict__ sv, const unsigned char *__restrict__ sv8,
                                             const double *sdict, const VT *__restrict__ x, double sx)
{
    int c[U];
    double a[U], xv[U];
    int code[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const size_t idx = e + (size_t)j * 64;
        c[j] = C16 ? cb + (int)ld_stream<NT>(&sc16[idx]) : ld_stream<NT>(&sc32[idx]);
        if (V8) code[j] = (int)ld_stream<NT>(&sv8[idx]);
        else a[j] = (double)ld_stream<NT>(&sv[idx]);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) xv[j] = (double)x[c[j]];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const double av = V8 ? sdict[code[j]] : a[j];
        const double p = av * (xv[j] * sx);
        if (k0 + j < len) sum = sum + p;
    }
    return sum;
}

static_assert(SELL_BLOCK == VEC_BLOCK, "the fused update runs k_update's blocks");

struct SellCoef {
    double sx, sy, cy;
};

// What every workgroup of a sliced-ELL product does before its rows: the lazy coefficients from
// the previous kernel's partial sums (spmv.h) and, with UPD, its share of the x/w update of the
// previous iteration (vec.h UpdArgs).  false = this product is skipped.
// In two halves: the coefficients ...
__device__ __forceinline__ bool sell_coefs(const SpmvCoef *__restrict__ coef, const double *__restrict__ pin, int npin,
                                           const NormSlot *__restrict__ slot_in, NormSlot *__restrict__ slot_out,
                                           int skip_if_zero, int wg, double *red, SellCoef &k, double &nrm_out,
                                           const NScale nsc, bool have_share = false, double share = 0.0)
{
    const int tid = threadIdx.x;
    if (pin != nullptr) {  // lazy coefficients (spmv.h); `share`: this thread's part of the partials, already summed
        const double nrm = sqrt(have_share ? block_sum_all_share<SELL_BLOCK>(share, red)
                                           : block_sum_all<SELL_BLOCK>(pin, npin, red)) * nsc.inv;
        nrm_out = nrm;
        if (skip_if_zero && !(nrm > 0.0)) {  // mode 2 is skipped when beta == 0 (:691)
            if (wg == 0 && tid == 0) {
                slot_out->nrm = nrm;
                slot_out->scale = 1.0;
            }
            return false;
        }
        k.sx = nrm > 0.0 ? 1.0 / nrm : 1.0;
        k.cy = -nrm;
        k.sy = slot_in->scale;
        if (wg == 0 && tid == 0) {
            slot_out->nrm = nrm;
            slot_out->scale = k.sx;
        }
        return true;
    }
    nrm_out = 0.0;
    if (coef->skip != 0) return false;
    k.sx = coef->sx;
    k.sy = coef->sy;
    k.cy = coef->cy;
    return true;
}
// ... and the update a lazy launch carries (nothing for an explicit-coefficient launch)
template <bool UPD, typename VT, bool NT = false>
__device__ __forceinline__ void sell_update(bool lazy, const NormSlot *__restrict__ slot_in, const UpdArgs &upd, int nwg,
                                            int wg, double *red, double nrm, double sx)
{
    if (!UPD || !lazy) return;
    const int tid = threadIdx.x;
    if (upd.on == 2) {  // the first launch of a solve: w <- v / alpha  (src/lsqr.f90:641-644)
        const XcdRange ur = xcd_range(upd.ugrid, nwg, wg);
        for (int ub = (int)ur.first; ub < (int)ur.end; ub += (int)ur.stride)
            winit_block<VT>((VT *)upd.w, (const VT *)upd.V, upd.n, sx, ub, upd.ugrid);
    } else if (upd.on) {  // x/w update of the previous iteration (vec.h UpdArgs)
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
            const double tot = update_block<VT, NT>((VT *)upd.x, (VT *)upd.w, (const VT *)upd.V, (VT *)upd.se, upd.n, rt.t1,
                                                rt.t2, rt.t3, sv, wantse, ub, upd.ugrid, red);
            if (tid == 0) upd.pout[ub] = tot;
        }
    }
}
template <bool UPD, typename VT, bool NT = false>
__device__ __forceinline__ bool sell_prologue(const SpmvCoef *__restrict__ coef, const double *__restrict__ pin, int npin,
                                              const NormSlot *__restrict__ slot_in, NormSlot *__restrict__ slot_out,
                                              int skip_if_zero, const UpdArgs &upd, int nwg, int wg, double *red,
                                              SellCoef &k, const NScale nsc, bool have_share = false, double share = 0.0)
{
    double nrm;
    if (!sell_coefs(coef, pin, npin, slot_in, slot_out, skip_if_zero, wg, red, k, nrm, nsc, have_share, share)) return false;
    sell_update<UPD, VT, NT>(pin != nullptr, slot_in, upd, nwg, wg, red, nrm, k.sx);
    return true;
}

// UPD = true: the launch also carries the x/w update of the previous iteration (UpdArgs).
template <bool C16, bool V8, bool UPD, typename VT = double, bool NT = false>
__global__ __launch_bounds__(SELL_BLOCK, 8) void k_spmv_sell(
    const unsigned *__restrict__ soff, const void *__restrict__ scolv, const int *__restrict__ cbaseS,
    const void *__restrict__ svalv, const double *__restrict__ dict, const unsigned char *__restrict__ rlen,
    int rows, int nslices, int64_t nblk, const VT *__restrict__ x, VT *__restrict__ y,
    const SpmvCoef *__restrict__ coef, const int *__restrict__ stop, double *__restrict__ partials,
    const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in,
    NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, UpdArgs upd, NScale nsc)
{
    __shared__ double red[SELL_BLOCK / WAVE + 1];
    __shared__ double sdict[V8 ? VD_MAX : 1];
    // block 0 carries the scalar rider (scalar.h), as in k_spmv_fused
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    // this thread's share of the previous kernel's partial sums, requested with the stop flag (as in pat.h: packed
    // records 40.3k -> 41.7k it/s at configs[1], 8-byte values 28.7k -> 29.2k; profiles/r03/config2_patterns.txt)
    const bool pre = pin != nullptr && npin <= SELL_SHARE_K * SELL_BLOCK;   // (uniform)
    double pshare[SELL_SHARE_K];
    if (pre) strided_share_load<SELL_BLOCK, SELL_SHARE_K>(pin, npin, pshare);
    if (*stop != 0) return;
    const int tid = threadIdx.x;
    if (V8) sdict[tid] = dict[tid];  // visible after the first barrier below

    SellCoef kc;
    const double share = pre ? strided_share_sum<SELL_BLOCK, SELL_SHARE_K>(pshare, npin) : 0.0;
    if (!sell_prologue<UPD, VT, NT>(coef, pin, npin, slot_in, slot_out, skip_if_zero, upd, nwg, wg, red, kc, nsc, pre, share))
        return;
    const double sx = kc.sx, sy = kc.sy, cy = kc.cy;
    if (V8) __syncthreads();

    const int *__restrict__ sc32 = static_cast<const int *>(scolv);
    const unsigned short *__restrict__ sc16 = static_cast<const unsigned short *>(scolv);
    const VT *__restrict__ sv = static_cast<const VT *>(svalv);
    const unsigned char *__restrict__ sv8 = static_cast<const unsigned char *>(svalv);
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: slice descriptors through the scalar cache

    double sq = 0.0;
    const XcdRange xr = xcd_range(nblk, nwg, wg);
    for (int64_t b = xr.first; b < xr.end; b += xr.stride) {
        const int s = (int)(b * SELL_SLICES) + wave;
        if (s >= nslices) continue;
        const unsigned o0 = (unsigned)__builtin_amdgcn_readfirstlane((int)soff[s]);
        const unsigned o1 = (unsigned)__builtin_amdgcn_readfirstlane((int)soff[s + 1]);
        const int W = (int)((o1 - o0) >> 6);
        const int cb = C16 ? __builtin_amdgcn_readfirstlane(cbaseS[s]) : 0;
        const int r = s * WAVE + lane;
        const bool active = r < rows;
        const int rc = active ? r : rows - 1;
        const int len = active ? (int)rlen[rc] : 0;
        const double y0 = (double)ld_stream<NT>(&y[rc]);
        double sum = 0.0;
        size_t e = (size_t)o0 + lane;
        int k0 = 0;
        for (; W - k0 >= 8; k0 += 8, e += 8 * 64)
            sum = sell_chunk<8, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx);
        switch (W - k0) {  // uniform
        case 7: sum = sell_chunk<7, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx); break;
        case 6: sum = sell_chunk<6, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx); break;
        case 5: sum = sell_chunk<5, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx); break;
        case 4: sum = sell_chunk<4, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx); break;
        case 3: sum = sell_chunk<3, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx); break;
        case 2: sum = sell_chunk<2, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx); break;
        case 1: sum = sell_chunk<1, C16, V8, VT, NT>(sum, e, k0, len, cb, sc32, sc16, sv, sv8, sdict, x, sx); break;
        default: break;
        }
        if (active) {
            const VT yn = (VT)(cy * (y0 * sy) + sum);
            y[r] = yn;
            const double ys = (double)yn * nsc.s;
            sq += ys * ys;
        }
    }
    const double tot = block_sum<SELL_BLOCK>(sq, red);
    if (tid == 0) partials[wg] = tot;
}

// ---------------------------------------------------------------------------------------------
// Packed records (sell = 2).  With 16-bit columns AND one-byte value codes a nonzero is 3 bytes,
// and the product above spends its time issuing loads, not moving bytes: per 64-row slice of a
// 5-point stencil 5 two-byte and 5 one-byte loads + the row lengths, each a wave instruction that
// fetches 64-128 B.  Here the same bytes come as ONE 16-byte load per lane:
//
//     record = 5 x u16 column (relative to the slice's smallest)  |  5 x u8 value code  |  u8 row length
//
// chunk j of slice s holds nonzeros 5j .. 5j+4 of its 64 rows: rec[roff[s] + 64 j + lane]; the row
// length is repeated in every chunk, rows of W nonzeros take ceil(W/5) records.  At config 2
// (W = 5) that is byte for byte the size of the unpacked layout; the product drops from 9.1 to
// 7.3 us (scripts/sell_roof.hip: 4.4 -> 5.5 TB/s, a pure stream of the same bytes takes 5.8).
// Chosen when the records cost <= 10 % more bytes than the unpacked slices (LSQRHIP_SELLP=0 never,
// =1 whenever columns and values qualify).  The sums are the same left-to-right sums.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sellp_fill(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const double *__restrict__ val, const unsigned *__restrict__ roff,
                                                    const int *__restrict__ cbaseS,
                                                    const unsigned long long *__restrict__ dict_bits, int nd, int rows,
                                                    int nslices, uint4 *__restrict__ rec)
{
    __shared__ unsigned long long tab[VD_MAX];
    if ((int)threadIdx.x < nd) tab[threadIdx.x] = dict_bits[threadIdx.x];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)(r >> 6), lane = (int)(r & 63);
    if (s >= nslices) return;
    const unsigned o0 = roff[s];
    const int nch = (int)((roff[s + 1] - o0) >> 6);
    const int cb = cbaseS[s];
    int q0 = 0, len = 0;
    if (r < rows) {
        q0 = rowptr[r];
        len = rowptr[r + 1] - q0;
    }
    for (int j = 0; j < nch; ++j) {
        unsigned c[SELLP_K], code[SELLP_K];
#pragma unroll
        for (int t = 0; t < SELLP_K; ++t) {
            const int k = SELLP_K * j + t;
            c[t] = 0;
            code[t] = 0;
            if (k < len) {
                c[t] = (unsigned)(col[q0 + k] - cb);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(val[q0 + k]);
                int lo = 0, hi = nd - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (tab[mid] < bits) lo = mid + 1;
                    else hi = mid;
                }
                code[t] = (unsigned)lo;
            }
        }
        uint4 q;
        q.x = c[0] | (c[1] << 16);
        q.y = c[2] | (c[3] << 16);
        q.z = c[4] | (code[0] << 16) | (code[1] << 24);
        q.w = code[2] | (code[3] << 8) | (code[4] << 16) | ((unsigned)len << 24);
        rec[(size_t)o0 + (size_t)j * 64 + lane] = q;
    }
}

// the 5 columns of one record
__device__ __forceinline__ void sellp_cols(const uint4 q, int cb, int (&c)[SELLP_K])
{
    c[0] = cb + (int)(q.x & 0xffffu);
    c[1] = cb + (int)(q.x >> 16);
    c[2] = cb + (int)(q.y & 0xffffu);
    c[3] = cb + (int)(q.y >> 16);
    c[4] = cb + (int)(q.z & 0xffffu);
}
// the 5 products of one record, added left to right where they exist (k0 = index of its first nonzero)
__device__ __forceinline__ double sellp_sum(double sum, const uint4 q, int k0, const double (&xv)[SELLP_K],
                                            const double *sdict, double sx)
{
    int code[SELLP_K];
    code[0] = (int)((q.z >> 16) & 0xffu);
    code[1] = (int)(q.z >> 24);
    code[2] = (int)(q.w & 0xffu);
    code[3] = (int)((q.w >> 8) & 0xffu);
    code[4] = (int)((q.w >> 16) & 0xffu);
    const int len = (int)(q.w >> 24);
#pragma unroll
    for (int t = 0; t < SELLP_K; ++t) {
        const double p = sdict[code[t]] * (xv[t] * sx);
        if (k0 + t < len) sum = sum + p;
    }
    return sum;
}
template <typename VT>
__device__ __forceinline__ double sellp_add(double sum, const uint4 q, int k0, int cb, const double *sdict,
                                            const VT *__restrict__ x, double sx)
{
    int c[SELLP_K];
    sellp_cols(q, cb, c);
    double xv[SELLP_K];
#pragma unroll
    for (int t = 0; t < SELLP_K; ++t) xv[t] = (double)x[c[t]];
    return sellp_sum(sum, q, k0, xv, sdict, sx);
}

// A launch at BASELINE configs[1] lives ~8 us, and what a workgroup does in it is a CHAIN of dependent round
// trips: stop flag -> dictionary -> the previous kernel's partial sums (the lazy norm: two trips and two
// barriers) -> slice descriptor -> record -> gathered x -> y.  Round 3 (profiles/r03/config2_early_loads.txt):
//   * the dictionary's load and the stop flag's are requested together at the top (one round trip instead of two)
//     and the dictionary is stored to LDS just before the prologue's barrier: kept, 39.5k -> 40.7k iterations/s
//     at K = 2000, 33.0k -> 33.8k at K = 20;
//   * ALSO requesting the first slice's record, y and gathered x before the prologue (so that they land while it
//     runs) makes the product alone 0.4 us faster and the solve 1 us per iteration SLOWER, whether they are
//     queued before or behind the prologue's own loads: 1536 workgroups' partial-sum loads then share the memory
//     system with 40 MB of product requests right behind a kernel boundary that is still writing the previous
//     kernel's output back, and the prologue -- which the fused update and every row sum wait for -- ends
//     later.  Not kept.
//   * (after pat.h) the slice descriptors two trips ahead, the first record of a slice one trip ahead -- in front of
//     or behind the current trip's gathers -- and this thread's share of the partial sums requested at the top: the
//     product alone 7.7 -> 8.5 us, the solve 40.2k -> 37.1k iterations/s (profiles/r03/config2_patterns.txt).  Not kept.
template <bool UPD, typename VT = double, bool NT = false>
__global__ __launch_bounds__(SELL_BLOCK, 6) void k_spmv_sellp(
    const unsigned *__restrict__ roff, const uint4 *__restrict__ rec, const int *__restrict__ cbaseS,
    const double *__restrict__ dict, int rows, int nslices, int64_t nblk, const VT *__restrict__ x,
    VT *__restrict__ y, const SpmvCoef *__restrict__ coef, const int *__restrict__ stop,
    double *__restrict__ partials, const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in,
    NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, UpdArgs upd, NScale nsc)
{
    __shared__ double red[SELL_BLOCK / WAVE + 1];
    __shared__ double sdict[VD_MAX];
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    const int tid = threadIdx.x;
    const double dict_mine = dict[tid];  // (requested here, stored below: one round trip with the stop flag's)
    const bool pre = pin != nullptr && npin <= SELL_SHARE_K * SELL_BLOCK;   // (uniform) ... and so is this thread's
    double pshare[SELL_SHARE_K];                                             // share of the partial sums
    if (pre) strided_share_load<SELL_BLOCK, SELL_SHARE_K>(pin, npin, pshare);
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the slice descriptors arrive in SGPRs
    const XcdRange xr = xcd_range(nblk, nwg, wg);

    if (*stop != 0) return;   // (its load shares the dictionary's round trip)
    sdict[tid] = dict_mine;   // visible after the barrier below

    SellCoef kc;
    const double share = pre ? strided_share_sum<SELL_BLOCK, SELL_SHARE_K>(pshare, npin) : 0.0;
    if (!sell_prologue<UPD, VT, NT>(coef, pin, npin, slot_in, slot_out, skip_if_zero, upd, nwg, wg, red, kc, nsc, pre, share))
        return;
    const double sx = kc.sx, sy = kc.sy, cy = kc.cy;
    __syncthreads();

    double sq = 0.0;
    for (int64_t b = xr.first; b < xr.end; b += xr.stride) {
        const int s = (int)(b * SELL_SLICES) + wave;
        if (s >= nslices) continue;
        const unsigned o0 = roff[s];
        const int nch = (int)((roff[s + 1] - o0) >> 6);
        const int cb = cbaseS[s];
        const int r = s * WAVE + lane;
        const bool active = r < rows;
        const double y0 = (double)ld_stream<NT>(&y[active ? r : rows - 1]);
        const uint4 *__restrict__ p = rec + (size_t)o0 + lane;
        double sum = 0.0;
        if (nch == 1) {
            sum = sellp_add<VT>(sum, ld_stream4<NT>(&p[0]), 0, cb, sdict, x, sx);
        } else {
            for (int j = 0; j < nch; j += 2) {  // two records in flight; the second one clamped, not branched on
                const uint4 qa = ld_stream4<NT>(&p[(size_t)j * 64]);
                const uint4 qb = ld_stream4<NT>(&p[(size_t)min(j + 1, nch - 1) * 64]);
                sum = sellp_add<VT>(sum, qa, SELLP_K * j, cb, sdict, x, sx);
                if (j + 1 < nch) sum = sellp_add<VT>(sum, qb, SELLP_K * (j + 1), cb, sdict, x, sx);
            }
        }
        if (active) {
            const VT yn = (VT)(cy * (y0 * sy) + sum);
            y[r] = yn;
            const double ys = (double)yn * nsc.s;
            sq += ys * ys;
        }
    }
    const double tot = block_sum<SELL_BLOCK>(sq, red);
    if (tid == 0) partials[wg] = tot;
}

}  // namespace lsqrhip
