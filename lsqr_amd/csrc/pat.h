// pat.h -- row patterns: SpMV for matrices whose rows repeat (aprod mode 1 / mode 2).
//
// Same contract as sell.h / spmv.h (reference src/lsqr.f90:166-174 / :186-194 fused with the dscal before and
// the dnrm2 after, :681-683 / :692-695):
//
//     y_i  <-  cy * (y_i * sy)  +  sum_j A_ij * (x_j * sx)        partial += y_i^2
//
// Why a third short-row layout.  A constant-coefficient stencil -- BASELINE.json configs[1], any finite-difference
// operator on a structured grid -- has a handful of DISTINCT rows: row i is (column - i, value) for a few offsets,
// the same list for every interior point, another few lists along the boundaries.  The packed records of sell.h
// already store such a row in 16 bytes; here it is ONE byte, the number of its pattern, and the patterns
// themselves (<= 256 lists, <= 1024 entries in all) sit in LDS.  A product then moves 1 + 8 + 16 = 25 bytes per
// row instead of 40 -- on a bandwidth-bound kernel the only thing that still helps.
//
//   pid[r]     = pattern of row r                                              (u8, the only per-row data)
//   desc[p]    = first entry of pattern p | its length << 16
//   delta[e], val[e] = entry e: column - row, value; the entries of a pattern in the row's COO order
//
// Lane i of a wave owns row i of a 64-row slice as in sell.h; its sum is the plain left-to-right sum over the
// pattern's entries starting from 0 -- bit-identical to the reference's and to the other short-row kernels.
// CONTRACT across the short-row layouts: every PRODUCT y is bit-equal in all of them (slice form, paired rows, sliced
// ELL, packed records).  The fused NORMS (sum of y_i^2) are bit-equal only between layouts with the same rows per
// thread -- the slice form and sell.h; the paired rows (the default) add the squares in another order, so alpha and beta
// may differ in the last bit and the iterates of a solve agree to rounding (<= 1e-12), not bit for bit
// (tests/test_gpu_patterns.py::test_paired_rows_and_the_slice_form_agree).
//
// Build (k_pat_* below, from the CSR of csr_build.h, all on the device): every row is hashed over (length, column
// offsets, value bits) into a 1024-slot table that gives up at the 257th distinct key; the keys are ranked (pattern
// numbers do not depend on the order the rows arrived in), each pattern is copied from its first row, and every row is
// then COMPARED with the pattern its hash names -- entry by entry, bit by bit; one mismatch (a hash collision) and the
// matrix keeps the layout it would have had.  A matrix with unrelated rows leaves after a fraction of one pass.
//
// Chosen when the matrix has <= 256 distinct rows of <= 64 nonzeros, <= 1024 pattern entries, and at least 16 rows per
// pattern.  LSQRHIP_PAT=0 never, =1 whenever the limits hold.
//
// PAIRED ROWS (round 5, the default: k_spmv_patp below): the same layout with lane L owning rows 2L, 2L + 1 of a 128-row
// group -- half the vector-memory requests per row, which is what these kernels run out of.  The slice form here
// (LSQRHIP_PAT_PAIR=0) keeps the rows-per-thread of sell.h, and with them norms bit-equal to the other short-row layouts.
//
// Second part of the file: STRUCTURE patterns (sell = 4) -- the same table over (length, column offsets) alone, for rows
// whose values do not repeat (variable coefficients): 8-byte values column-major per slice and no column indices.
// Third part: WIDE row patterns -- 257 ... 4096 distinct rows: two bytes per row, the table in global memory.
#pragma once

#include "common.h"
#include "scalar.h"
#include "sell.h"
#include "state.h"
#include "vec.h"

namespace lsqrhip {

constexpr int PAT_MAX = 256;       // patterns (one byte per row)
constexpr int PAT_MAX_E = 1024;    // entries of all patterns together (12 KB of LDS)
// k_spmv_pat / k_spmv_spat stage the table with one descriptor per THREAD (desc[tid]) and PAT_MAX_E / SELL_BLOCK entries
// per thread: the constants are tied to the workgroup size of sell.h
static_assert(SELL_BLOCK == PAT_MAX, "one pattern descriptor per thread of a SELL_BLOCK workgroup");
static_assert(PAT_MAX_E % SELL_BLOCK == 0, "the pattern entries divide evenly among the threads");
constexpr int PAT_MAX_LEN = 64;    // nonzeros of a row
constexpr int PAT_TAB = 1024;      // slots of the discovery table
constexpr int PAT_K = 5;           // entries in flight per lane and slice
constexpr int PAT_U = 2;           // slices a wave takes through the chain together
constexpr int PAT_SHARE_K = 4;     // partial sums of the previous kernel per thread requested at the top (<= 1024 of them)
constexpr int PAT_MAX_GRID = 1024; // 4 workgroups per CU x 256 CUs (lsqrhip.hip)

__device__ __forceinline__ unsigned long long pat_mix(unsigned long long h, unsigned long long v)
{
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    return h;
}
// key of row r (never 0: 0 marks an empty slot)
// (vals = 0: the key of the row's STRUCTURE alone -- length and column offsets -- for the structure patterns below)
__device__ __forceinline__ unsigned long long pat_row_key(const int *__restrict__ col, const double *__restrict__ val,
                                                          int q0, int len, int r, int vals)
{
    unsigned long long h = pat_mix(0x243f6a8885a308d3ull, (unsigned long long)len);
    for (int k = 0; k < len; ++k) {
        h = pat_mix(h, (unsigned long long)(unsigned)(col[q0 + k] - r));
        if (vals) h = pat_mix(h, (unsigned long long)__double_as_longlong(val[q0 + k]));
    }
    return h | 1ull;
}

// ctl[0] = distinct keys so far, ctl[1] = give up, ctl[2] = patterns, ctl[3] = entries (k_pat_table)
__global__ __launch_bounds__(256) void k_pat_discover(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      const double *__restrict__ val, int rows, int vals,
                                                      unsigned long long *__restrict__ keys, int *__restrict__ reps,
                                                      int *__restrict__ ctl, int tab = PAT_TAB, int pmax = PAT_MAX)
{
    // (tab slots, a power of two; give up at the (pmax + 1)-th distinct key: PAT_TAB / PAT_MAX, or PAT2_TAB / PAT2_MAX for
    // the wide table below)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        if (*(volatile int *)&ctl[1] != 0) return;
        const int q0 = rowptr[r], len = rowptr[r + 1] - q0;
        if (len > PAT_MAX_LEN) {
            ctl[1] = 1;
            return;
        }
        const unsigned long long h = pat_row_key(col, val, q0, len, (int)r, vals);
        unsigned slot = (unsigned)(h >> 11) & (unsigned)(tab - 1);
        int tries = 0;
        for (; tries < tab; ++tries) {
            // look before the exchange: a million interior rows of a stencil share ONE key, and a million compare-and-
            // swaps on one word take 11 ms; a device-scope load (the L2 of another XCD may hold the word as it was) does not
            unsigned long long old = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == 0ull) old = atomicCAS(&keys[slot], 0ull, h);
            if (old == 0ull) {
                if (atomicAdd(&ctl[0], 1) >= pmax) ctl[1] = 1;
                atomicMin(&reps[slot], (int)r);
                break;
            }
            if (old == h) {
                if (__hip_atomic_load(&reps[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (int)r)
                    atomicMin(&reps[slot], (int)r);
                break;
            }
            if (*(volatile int *)&ctl[1] != 0) return;
            slot = (slot + 1) & (unsigned)(tab - 1);
        }
        if (tries == tab) {
            ctl[1] = 1;
            return;
        }
    }
}

// One workgroup of PAT_TAB threads: rank the keys (pattern p = p-th smallest key), lay the patterns out, copy each from
// its first row.  slot_pat[slot] = pattern of the key in that slot.
__global__ __launch_bounds__(PAT_TAB) void k_pat_table(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                       const double *__restrict__ val, int vals,
                                                       const unsigned long long *__restrict__ keys,
                                                       const int *__restrict__ reps, int *__restrict__ slot_pat,
                                                       unsigned *__restrict__ desc, int *__restrict__ delta,
                                                       double *__restrict__ pval, int *__restrict__ ctl)
{
    __shared__ unsigned long long skey[PAT_TAB];
    __shared__ int slen[PAT_MAX], sstart[PAT_MAX + 1], srep[PAT_MAX];
    const int t = threadIdx.x;
    if (ctl[1] != 0) return;
    const unsigned long long key = keys[t];
    skey[t] = key;
    if (t < PAT_MAX) slen[t] = 0;
    __syncthreads();
    int p = -1;
    if (key != 0ull) {
        p = 0;
        for (int j = 0; j < PAT_TAB; ++j) p += (skey[j] != 0ull && skey[j] < key) ? 1 : 0;
    }
    slot_pat[t] = p;
    if (p >= 0 && p < PAT_MAX) {
        const int rep = reps[t];
        srep[p] = rep;
        slen[p] = rowptr[rep + 1] - rowptr[rep];
    }
    __syncthreads();
    const int np = min(ctl[0], PAT_MAX);
    if (t == 0) {
        int e = 0;
        for (int j = 0; j < np; ++j) {
            sstart[j] = e;
            e += slen[j];
        }
        sstart[np] = e;
        ctl[2] = np;
        ctl[3] = e;
        if (e > PAT_MAX_E) ctl[1] = 1;
    }
    __syncthreads();
    if (sstart[np] > PAT_MAX_E) return;
    if (t < PAT_MAX) desc[t] = t < np ? ((unsigned)sstart[t] | ((unsigned)slen[t] << 16)) : 0u;
    if (t < np) {
        const int rep = srep[t], q0 = rowptr[rep], e0 = sstart[t];
        for (int k = 0; k < slen[t]; ++k) {
            delta[e0 + k] = col[q0 + k] - rep;
            if (vals) pval[e0 + k] = val[q0 + k];
        }
    }
}

// pid[r] = pattern of row r, after comparing the row with it entry by entry (ctl[1] = 1 on any difference)
__global__ __launch_bounds__(256) void k_pat_assign(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    const double *__restrict__ val, int rows, int vals,
                                                    const unsigned long long *__restrict__ keys,
                                                    const int *__restrict__ slot_pat, const unsigned *__restrict__ desc,
                                                    const int *__restrict__ delta, const double *__restrict__ pval,
                                                    unsigned char *__restrict__ pid, int *__restrict__ ctl)
{
    // (k_pat_table gave up -- more entries than the table holds -- and left desc / delta unwritten: nothing to compare with)
    if (*(volatile int *)&ctl[1] != 0) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += stride) {
        const int q0 = rowptr[r], len = rowptr[r + 1] - q0;
        const unsigned long long h = pat_row_key(col, val, q0, len, (int)r, vals);
        unsigned slot = (unsigned)(h >> 11) & (PAT_TAB - 1);
        int p = -1;
        for (int tries = 0; tries < PAT_TAB; ++tries) {
            const unsigned long long k = keys[slot];
            if (k == h) {
                p = slot_pat[slot];
                break;
            }
            if (k == 0ull) break;
            slot = (slot + 1) & (PAT_TAB - 1);
        }
        bool same = p >= 0 && p < PAT_MAX;
        if (same) {
            const unsigned d = desc[p];
            const int e0 = (int)(d & 0xffffu);
            same = (int)(d >> 16) == len && e0 + len <= PAT_MAX_E;
            for (int k = 0; same && k < len; ++k)
                same = delta[e0 + k] == col[q0 + k] - (int)r &&
                       (!vals || __double_as_longlong(pval[e0 + k]) == __double_as_longlong(val[q0 + k]));
        }
        if (!same) {
            ctl[1] = 1;
            return;
        }
        pid[r] = (unsigned char)p;
    }
}

// The rows of PAT_U slices on their way through one trip: pattern -> entries (LDS) -> gathered x, y.
struct PatTrip {
    int r[PAT_U], e0[PAT_U], len[PAT_U];
    bool active[PAT_U];
    double y0[PAT_U];
    double a[PAT_U][PAT_K], xv[PAT_U][PAT_K];
};
// request everything the first PAT_K entries of the rows need
template <typename VT, bool NT>
__device__ __forceinline__ void pat_issue(PatTrip &t, int64_t b, const XcdRange &xr, int wave, int lane, int rows,
                                          const int (&pidc)[PAT_U], const unsigned *sdesc, const int *sdelta,
                                          const double *sval, const VT *__restrict__ x, const VT *y)
{
#pragma unroll
    for (int u = 0; u < PAT_U; ++u) {
        const int64_t bu = b + u * xr.stride;
        const int64_t r64 = (bu * SELL_SLICES + wave) * WAVE + lane;   // (64-bit: bu beyond xr.end may pass 2^31 rows)
        t.active[u] = bu < xr.end && r64 < rows;
        t.r[u] = t.active[u] ? (int)r64 : 0;
        t.y0[u] = (double)ld_stream<NT>(&y[t.active[u] ? t.r[u] : 0]);
        const unsigned d = t.active[u] ? sdesc[pidc[u]] : 0u;
        t.e0[u] = (int)(d & 0xffffu);
        t.len[u] = (int)(d >> 16);
    }
#pragma unroll
    for (int u = 0; u < PAT_U; ++u)
#pragma unroll
        for (int k = 0; k < PAT_K; ++k) {
            const bool live = k < t.len[u];
            const int e = live ? t.e0[u] + k : 0;
            t.a[u][k] = sval[e];
            t.xv[u][k] = (double)x[live ? t.r[u] + sdelta[e] : 0];   // (no entry: x[0], never added)
        }
}
// the left-to-right sums, the rest of rows longer than PAT_K, y and its square
template <typename VT>
__device__ __forceinline__ void pat_finish(const PatTrip &t, double sx, double sy, double cy, const NScale nsc,
                                           const int *sdelta, const double *sval, const VT *__restrict__ x,
                                           VT *__restrict__ y, double &sq)
{
    double sum[PAT_U];
#pragma unroll
    for (int u = 0; u < PAT_U; ++u) {
        sum[u] = 0.0;
#pragma unroll
        for (int k = 0; k < PAT_K; ++k) {
            const double p = t.a[u][k] * (t.xv[u][k] * sx);
            if (k < t.len[u]) sum[u] = sum[u] + p;
        }
    }
#pragma unroll
    for (int u = 0; u < PAT_U; ++u) {
        for (int k0 = PAT_K; __any(k0 < t.len[u]); k0 += PAT_K) {
            double a[PAT_K], xv[PAT_K];
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                const bool live = k0 + k < t.len[u];
                const int e = live ? t.e0[u] + k0 + k : 0;
                a[k] = sval[e];
                xv[k] = (double)x[live ? t.r[u] + sdelta[e] : 0];
            }
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                const double p = a[k] * (xv[k] * sx);
                if (k0 + k < t.len[u]) sum[u] = sum[u] + p;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PAT_U; ++u) {
        if (t.active[u]) {
            const VT yn = (VT)(cy * (t.y0[u] * sy) + sum[u]);
            store_through(&y[t.r[u]], yn);
            const double ys = (double)yn * nsc.s;
            sq += ys * ys;
        }
    }
}

// UPD = true: the launch also carries the x/w update of the previous iteration (UpdArgs), as in sell.h.
//
// At configs[1] a wave has four slices and the launch lives ~7 us: the CHAIN of dependent round trips, not the bytes,
// is what it waits for (stop flag, table -> the previous kernel's partial sums -> pattern number -> gathered x -> y).
// So: the table, the stop flag and the pattern numbers of the first trip (a byte per lane) are requested at the top in
// one go; PAT_U slices go through a trip together; the pattern numbers of the next trip are requested before the
// gathers of this one; this thread's share of the previous kernel's partial sums is requested at the top as well
// (46.6k -> 47.6k iterations/s).  NOT: the first trip's gathers ahead of the prologue (the product alone takes the
// same 7.2 us, the solve drops from 46.7k to 45.5k iterations/s, as it did for the packed records of sell.h), nor
// between the prologue's norm and the update it carries (47.5k -> 44.3k): requests in flight while the update
// streams delay it more than they save (profiles/r03/config2_patterns.txt).
template <bool UPD, typename VT = double, bool NT = false>
__global__ __launch_bounds__(SELL_BLOCK, 4) void k_spmv_pat(
    const unsigned char *__restrict__ pid, const unsigned *__restrict__ desc, const int *__restrict__ delta,
    const double *__restrict__ pval, int nent, int rows, int nslices, int64_t nblk, const VT *__restrict__ x,
    VT *__restrict__ y, const SpmvCoef *__restrict__ coef, const int *__restrict__ stop,
    double *__restrict__ partials, const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in,
    NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, UpdArgs upd, NScale nsc)
{
    __shared__ double red[SELL_BLOCK / WAVE + 1];
    __shared__ unsigned sdesc[PAT_MAX];
    __shared__ int sdelta[PAT_MAX_E];
    __shared__ double sval[PAT_MAX_E];
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    const int tid = threadIdx.x;
    const unsigned desc_mine = desc[tid];
    int d_mine[PAT_MAX_E / SELL_BLOCK];
    double v_mine[PAT_MAX_E / SELL_BLOCK];
#pragma unroll
    for (int j = 0; j < PAT_MAX_E / SELL_BLOCK; ++j) {
        const int e = tid + j * SELL_BLOCK;
        d_mine[j] = e < nent ? delta[e] : 0;
        v_mine[j] = e < nent ? pval[e] : 0.0;
    }
    // ... and this thread's share of the previous kernel's partial sums (the lazy norm of the prologue)
    const bool pre = pin != nullptr && npin <= PAT_SHARE_K * SELL_BLOCK;   // (uniform)
    double pshare[PAT_SHARE_K];
    if (pre) strided_share_load<SELL_BLOCK, PAT_SHARE_K>(pin, npin, pshare);
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdRange xr = xcd_range(nblk, nwg, wg);
    int64_t b = xr.first;
    int pid_next[PAT_U];
#pragma unroll
    for (int u = 0; u < PAT_U; ++u) {
        const int64_t bu = b + u * xr.stride;
        const int64_t r0 = (bu * SELL_SLICES + wave) * WAVE + lane;
        pid_next[u] = (bu < xr.end && r0 < rows) ? (int)ld_stream<NT>(&pid[r0]) : 0;
    }

    if (*stop != 0) return;
    sdesc[tid] = desc_mine;
#pragma unroll
    for (int j = 0; j < PAT_MAX_E / SELL_BLOCK; ++j) {
        const int e = tid + j * SELL_BLOCK;
        if (e < nent) {
            sdelta[e] = d_mine[j];
            sval[e] = v_mine[j];
        }
    }
    SellCoef kc;
    const double share = pre ? strided_share_sum<SELL_BLOCK, PAT_SHARE_K>(pshare, npin) : 0.0;
    if (!sell_prologue<UPD, VT, NT>(coef, pin, npin, slot_in, slot_out, skip_if_zero, upd, nwg, wg, red, kc, nsc, pre, share))
        return;
    const double sx = kc.sx, sy = kc.sy, cy = kc.cy;
    __syncthreads();

    double sq = 0.0;
    for (; b < xr.end; b += PAT_U * xr.stride) {
        int pidc[PAT_U];
#pragma unroll
        for (int u = 0; u < PAT_U; ++u) pidc[u] = pid_next[u];
#pragma unroll
        for (int u = 0; u < PAT_U; ++u) {
            const int64_t bn = b + (PAT_U + u) * xr.stride;
            const int rn = ((int)(bn * SELL_SLICES) + wave) * WAVE + lane;
            pid_next[u] = (bn < xr.end && rn < rows) ? (int)ld_stream<NT>(&pid[rn]) : 0;
        }
        PatTrip trip;
        pat_issue<VT, NT>(trip, b, xr, wave, lane, rows, pidc, sdesc, sdelta, sval, x, y);
        pat_finish<VT>(trip, sx, sy, cy, nsc, sdelta, sval, x, y, sq);
    }
    const double tot = block_sum<SELL_BLOCK>(sq, red);
    if (tid == 0) partials[wg] = tot;
}

// ---------------------------------------------------------------------------------------------------------------
// PAIRED ROWS (Csr.pat_pair): the same one-byte layout, the same table in LDS, the same row sums -- with lane L of a
// wave owning rows 2L and 2L + 1 of a 128-row group instead of row L of a 64-row slice.  What the pattern kernels run
// out of is vector-memory REQUESTS, not bytes (profiles/r05/pmc_short_rows.txt: k_spmv_pat issues 7.1 reads + 1 store per
// 64 rows at ~26 cycles each per CU, the texture addressers 70 % busy, at 0.59 of the HBM peak).  Two neighbouring rows
// per lane turn every stream into 16-byte requests -- two pattern numbers in one 2-byte load, y read and stored as a
// pair -- and, where the two rows name the SAME pattern (every lane of a group: wave-uniform; a stencil's interior), the
// two gathers of an entry into ONE 16-byte gather of x[r + delta], x[r + 1 + delta] (8-byte aligned): 8 requests per
// 128 rows where the slice kernel has 16.  A group with a lane whose rows differ -- the ends of a grid row -- gathers
// per row, as before.  Every row sum is the same left-to-right sum; the partial sums of the norms run over other
// rows per thread, so a norm may differ from the slice kernels' in its last bit.
// Measured (profiles/r05/pair_ab.txt, one process per pair of lines): 16M-row Poisson 85-86 -> 71-72 us per product
// (0.59 -> 0.70 of 8 TB/s on its 25 bytes per row); configs[1] 7.3 -> 6.7 us per product, 47.5-47.8k -> 49.2-49.4k
// iterations/s at K = 2000 and 41.0-41.5k -> 42.4-43.1k at K = 20.  Two groups per trip (128 registers): 45.5-46.6k.
// 1024 workgroups stay the best grid (768: 46.5-47.0k, 1536: 45.4k, 2048: 38.7k; profiles/r05/pair_grid.txt).
// Measured and dropped (profiles/r05/pair_mid_ab.txt): the middle entry of a run of columns d - 1, d, d + 1 -- the -1, 0,
// +1 of a stencil -- formed in the lane from the pairs before and behind it instead of gathered (7 requests per 128
// rows instead of 8): the wave-uniform test in front of every gather breaks the batch of requests up -- 70 -> 85 us
// at 16M rows, 49.7k -> 43.4k it/s at configs[1].
// ---------------------------------------------------------------------------------------------------------------
typedef double lsqrhip_d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef float lsqrhip_f2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void ld_pair(const double *p, double &a, double &b)
{
    const lsqrhip_d2u v = *reinterpret_cast<const lsqrhip_d2u *>(p);
    a = v.x;
    b = v.y;
}
__device__ __forceinline__ void ld_pair(const float *p, double &a, double &b)
{
    const lsqrhip_f2u v = *reinterpret_cast<const lsqrhip_f2u *>(p);
    a = (double)v.x;
    b = (double)v.y;
}
__device__ __forceinline__ void store_through2(double *p, double a, double b)
{
    lsqrhip_d2 v;
    v.x = a;
    v.y = b;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_through2(float *p, float a, float b)
{
    lsqrhip_f2 v;
    v.x = a;
    v.y = b;
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <bool UPD, typename VT = double, bool NT = false, int U = 1>
__global__ __launch_bounds__(SELL_BLOCK, 4) void k_spmv_patp(
    const unsigned char *__restrict__ pid, const unsigned *__restrict__ desc, const int *__restrict__ delta,
    const double *__restrict__ pval, int nent, int rows, int64_t nblk, const VT *__restrict__ x, VT *__restrict__ y,
    const SpmvCoef *__restrict__ coef, const int *__restrict__ stop, double *__restrict__ partials,
    const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in, NormSlot *__restrict__ slot_out,
    int skip_if_zero, Rider rider, UpdArgs upd, NScale nsc)
{
    typedef typename Vec2<VT>::type V2T;
    __shared__ double red[SELL_BLOCK / WAVE + 1];
    __shared__ unsigned sdesc[PAT_MAX];
    __shared__ int sdelta[PAT_MAX_E];
    __shared__ double sval[PAT_MAX_E];
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    const int tid = threadIdx.x;
    const unsigned desc_mine = desc[tid];
    int d_mine[PAT_MAX_E / SELL_BLOCK];
    double v_mine[PAT_MAX_E / SELL_BLOCK];
#pragma unroll
    for (int j = 0; j < PAT_MAX_E / SELL_BLOCK; ++j) {
        const int e = tid + j * SELL_BLOCK;
        d_mine[j] = e < nent ? delta[e] : 0;
        v_mine[j] = e < nent ? pval[e] : 0.0;
    }
    const bool pre = pin != nullptr && npin <= PAT_SHARE_K * SELL_BLOCK;   // (uniform)
    double pshare[PAT_SHARE_K];
    if (pre) strided_share_load<SELL_BLOCK, PAT_SHARE_K>(pin, npin, pshare);
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdRange xr = xcd_range(nblk, nwg, wg);   // blocks of SELL_SLICES groups of 2 * WAVE rows
    // the two pattern numbers of the lane's rows in group (b, wave) (pid has two bytes of padding behind the last row)
    auto pids_of = [&](int64_t b) -> int {
        const int64_t r0 = (b * SELL_SLICES + wave) * (2 * WAVE) + 2 * lane;
        return (b < xr.end && r0 < rows) ? (int)ld_stream<NT>(reinterpret_cast<const unsigned short *>(pid + r0)) : 0;
    };
    int64_t b = xr.first;
    int pid_next[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pid_next[u] = pids_of(b + u * xr.stride);

    if (*stop != 0) return;
    sdesc[tid] = desc_mine;
#pragma unroll
    for (int j = 0; j < PAT_MAX_E / SELL_BLOCK; ++j) {
        const int e = tid + j * SELL_BLOCK;
        if (e < nent) {
            sdelta[e] = d_mine[j];
            sval[e] = v_mine[j];
        }
    }
    SellCoef kc;
    const double share = pre ? strided_share_sum<SELL_BLOCK, PAT_SHARE_K>(pshare, npin) : 0.0;
    if (!sell_prologue<UPD, VT, NT>(coef, pin, npin, slot_in, slot_out, skip_if_zero, upd, nwg, wg, red, kc, nsc, pre, share))
        return;
    const double sx = kc.sx, sy = kc.sy, cy = kc.cy;
    __syncthreads();

    double sq = 0.0;
    for (; b < xr.end; b += U * xr.stride) {
        int r0[U], e0[U], len0[U], e1[U], len1[U];
        bool act0[U], act1[U], full[U];
        double y0[U], y1[U], a0[U][PAT_K], a1[U][PAT_K], x0[U][PAT_K], x1[U][PAT_K];
        int pp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pp[u] = pid_next[u];
#pragma unroll
        for (int u = 0; u < U; ++u) pid_next[u] = pids_of(b + (U + u) * xr.stride);
        // everything the first PAT_K entries of the rows of the trip's groups need, requested before any of it is used
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t bu = b + u * xr.stride;
            const int64_t R0 = (bu * SELL_SLICES + wave) * (2 * WAVE);
            const int64_t r64 = R0 + 2 * lane;
            const bool in = bu < xr.end;
            act0[u] = in && r64 < rows;
            act1[u] = in && r64 + 1 < rows;
            r0[u] = act0[u] ? (int)r64 : 0;
            const unsigned d0 = act0[u] ? sdesc[pp[u] & 255] : 0u, d1 = act1[u] ? sdesc[pp[u] >> 8] : 0u;
            e0[u] = (int)(d0 & 0xffffu);
            len0[u] = (int)(d0 >> 16);
            e1[u] = (int)(d1 & 0xffffu);
            len1[u] = (int)(d1 >> 16);
            full[u] = in && R0 + 2 * WAVE <= rows;                                              // (uniform)
            const bool pair = full[u] && __all((pp[u] & 255) == (pp[u] >> 8)) != 0;            // (uniform) every lane: one pattern for both rows
            if (full[u]) {
                const V2T yv = ld_stream2<NT>(reinterpret_cast<const V2T *>(&y[r0[u]]));
                y0[u] = (double)yv.x;
                y1[u] = (double)yv.y;
            } else {
                y0[u] = act0[u] ? (double)y[r0[u]] : 0.0;
                y1[u] = act1[u] ? (double)y[r0[u] + 1] : 0.0;
            }
            if (pair) {
#pragma unroll
                for (int k = 0; k < PAT_K; ++k) {
                    const bool live = k < len0[u];
                    const int e = live ? e0[u] + k : 0;
                    a0[u][k] = sval[e];
                    a1[u][k] = a0[u][k];
                    ld_pair(&x[live ? r0[u] + sdelta[e] : 0], x0[u][k], x1[u][k]);   // (no entry: x[0], x[1], never added; the layout needs two columns)
                }
            } else {
#pragma unroll
                for (int k = 0; k < PAT_K; ++k) {
                    const bool l0 = k < len0[u], l1 = k < len1[u];
                    const int ea = l0 ? e0[u] + k : 0, eb = l1 ? e1[u] + k : 0;
                    a0[u][k] = sval[ea];
                    a1[u][k] = sval[eb];
                    x0[u][k] = (double)x[l0 ? r0[u] + sdelta[ea] : 0];
                    x1[u][k] = (double)x[l1 ? r0[u] + 1 + sdelta[eb] : 0];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double sum0 = 0.0, sum1 = 0.0;
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                const double p0 = a0[u][k] * (x0[u][k] * sx), p1 = a1[u][k] * (x1[u][k] * sx);
                if (k < len0[u]) sum0 = sum0 + p0;
                if (k < len1[u]) sum1 = sum1 + p1;
            }
            for (int k0 = PAT_K; __any(k0 < len0[u] || k0 < len1[u]); k0 += PAT_K) {   // rows of more than PAT_K nonzeros
                double b0[PAT_K], b1[PAT_K], z0[PAT_K], z1[PAT_K];
#pragma unroll
                for (int k = 0; k < PAT_K; ++k) {
                    const bool l0 = k0 + k < len0[u], l1 = k0 + k < len1[u];
                    const int ea = l0 ? e0[u] + k0 + k : 0, eb = l1 ? e1[u] + k0 + k : 0;
                    b0[k] = sval[ea];
                    b1[k] = sval[eb];
                    z0[k] = (double)x[l0 ? r0[u] + sdelta[ea] : 0];
                    z1[k] = (double)x[l1 ? r0[u] + 1 + sdelta[eb] : 0];
                }
#pragma unroll
                for (int k = 0; k < PAT_K; ++k) {
                    const double p0 = b0[k] * (z0[k] * sx), p1 = b1[k] * (z1[k] * sx);
                    if (k0 + k < len0[u]) sum0 = sum0 + p0;
                    if (k0 + k < len1[u]) sum1 = sum1 + p1;
                }
            }
            const VT yn0 = (VT)(cy * (y0[u] * sy) + sum0), yn1 = (VT)(cy * (y1[u] * sy) + sum1);
            if (full[u]) {
                store_through2(&y[r0[u]], yn0, yn1);
            } else {
                if (act0[u]) store_through(&y[r0[u]], yn0);
                if (act1[u]) store_through(&y[r0[u] + 1], yn1);
            }
            if (act0[u]) {
                const double ys = (double)yn0 * nsc.s;
                sq += ys * ys;
            }
            if (act1[u]) {
                const double ys = (double)yn1 * nsc.s;
                sq += ys * ys;
            }
        }
    }
    const double tot = block_sum<SELL_BLOCK>(sq, red);
    if (tid == 0) partials[wg] = tot;
}

// ---------------------------------------------------------------------------------------------------------------
// Structure patterns (sell = 4): rows whose column STRUCTURE repeats while their values do not -- a stencil with
// variable coefficients.  The pattern table holds (length, column offsets) only; the values stay 8 bytes each,
// column-major per 64-row slice as in sell.h (element (row i of slice s, k) at soff[s] + 64 k + i, padding 0.0, never
// added).  Against sliced ELL with 16-bit columns that is 2 bytes less per nonzero and a byte per row instead of the
// row length + slice base: 65 instead of 75 bytes per row of a 5-point operator.  Same left-to-right row sums.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_spat_fill(const int *__restrict__ rowptr, const double *__restrict__ val,
                                                   const unsigned *__restrict__ soff, int rows, int nslices,
                                                   double *__restrict__ sval)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)(r >> 6), lane = (int)(r & 63);
    if (s >= nslices) return;
    const unsigned o0 = soff[s];
    const int W = (int)((soff[s + 1] - o0) >> 6);
    int q0 = 0, len = 0;
    if (r < rows) {
        q0 = rowptr[r];
        len = rowptr[r + 1] - q0;
    }
    for (int k = 0; k < W; ++k) sval[(size_t)o0 + (size_t)k * 64 + lane] = k < len ? val[q0 + k] : 0.0;
}

template <bool UPD, typename VT = double, bool NT = false>
__global__ __launch_bounds__(SELL_BLOCK, 6) void k_spmv_spat(
    const unsigned char *__restrict__ pid, const unsigned *__restrict__ desc, const int *__restrict__ delta, int nent,
    const unsigned *__restrict__ soff, const VT *__restrict__ sv, int rows, int nslices, int64_t nblk,
    const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef, const int *__restrict__ stop,
    double *__restrict__ partials, const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in,
    NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, UpdArgs upd, NScale nsc)
{
    __shared__ double red[SELL_BLOCK / WAVE + 1];
    __shared__ unsigned sdesc[PAT_MAX];
    __shared__ int sdelta[PAT_MAX_E];
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    const int tid = threadIdx.x;
    const unsigned desc_mine = desc[tid];
    int d_mine[PAT_MAX_E / SELL_BLOCK];
#pragma unroll
    for (int j = 0; j < PAT_MAX_E / SELL_BLOCK; ++j) {
        const int e = tid + j * SELL_BLOCK;
        d_mine[j] = e < nent ? delta[e] : 0;
    }
    const bool pre = pin != nullptr && npin <= SELL_SHARE_K * SELL_BLOCK;   // (uniform)
    double pshare[SELL_SHARE_K];
    if (pre) strided_share_load<SELL_BLOCK, SELL_SHARE_K>(pin, npin, pshare);
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdRange xr = xcd_range(nblk, nwg, wg);

    if (*stop != 0) return;
    sdesc[tid] = desc_mine;
#pragma unroll
    for (int j = 0; j < PAT_MAX_E / SELL_BLOCK; ++j) {
        const int e = tid + j * SELL_BLOCK;
        if (e < nent) sdelta[e] = d_mine[j];
    }
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
        const unsigned o0 = soff[s];
        const int W = (int)((soff[s + 1] - o0) >> 6);
        const int r = s * WAVE + lane;
        const bool active = r < rows;
        const double y0 = (double)ld_stream<NT>(&y[active ? r : 0]);
        const unsigned d = active ? sdesc[ld_stream<NT>(&pid[r])] : 0u;
        const int e0 = (int)(d & 0xffffu), len = (int)(d >> 16);
        const VT *__restrict__ pv = sv + (size_t)o0 + lane;
        double sum = 0.0;
        for (int k0 = 0; k0 < W; k0 += PAT_K) {   // W is the slice's longest row (uniform)
            double a[PAT_K], xv[PAT_K];
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                const bool live = k0 + k < len;
                a[k] = (double)ld_stream<NT>(&pv[(size_t)min(k0 + k, W - 1) * 64]);
                xv[k] = (double)x[live ? r + sdelta[e0 + k0 + k] : 0];
            }
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                const double p = a[k] * (xv[k] * sx);
                if (k0 + k < len) sum = sum + p;
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

// ---------------------------------------------------------------------------------------------------------------
// WIDE row patterns (sell = 3, Csr.pat_wide): 257 ... 4096 distinct rows -- a stencil whose coefficients are constant
// on each of many regions (materials, refinement levels), every region with its own interior and interface rows.
// The pattern number of a row takes two bytes and the table no longer fits the LDS of four workgroups per CU, so it
// stays in global memory and is read through L1 / L2 (a megabyte at most, and the rows of one slice name the same few
// patterns): a product moves 2 + 8 + 16 = 26 bytes per row, against the 65 of structure patterns.
//
//   pid[r]             = pattern of row r                                                      (u16)
//   ent[p * stride + k] = { value, column - row, length of the pattern } of entry k of pattern p, 16 bytes: ONE request
//                         per entry where the LDS table has two reads, and no descriptor to fetch first -- stride = the
//                         longest pattern (>= 1), the length rides in every entry
//
// Same rows, same left-to-right sums from 0, the same contract: every bit of a product equals the other short-row
// layouts'.
//
// What the kernel is short of is vector-memory INSTRUCTIONS, not bytes and not latency (16M-row mesh of 2304 patterns,
// profiles/r05/wide_patterns.txt): with a descriptor and five entry requests per 64-row slice beside the eight of
// k_spmv_pat a product took 124 us whatever the occupancy (4, 6, 8 workgroups per CU), however many slices went
// through a trip together, with the entries requested a trip ahead (117 us), and with the entries requested by 8 of the
// 64 lanes only (120 us: a gather costs the same whoever takes part); without the entry requests 90 us.  So:
//   * no descriptor (the fixed stride above);
//   * a slice whose rows all name ONE pattern -- the interior of a region, three slices in four on that mesh -- asks
//     for its entries with ONE request, lane k for entry k, and hands them round with v_readlane; only a slice that
//     crosses an interface requests entry k of every lane's own pattern, PAT_K requests (every slice that way: 127 us);
//   * the trip is a software pipeline: pattern numbers of trip t + 2, entries of t + 1, gathered x of t.
// Together 103-104 us on that mesh (0.50 of 8 TB/s on its 26 bytes per row; structure patterns, what the matrix had
// before: 175 us; the one-byte kernel on a mesh of 9 patterns: 84 us), the same at 4 to 16 workgroups per CU.
// Measured and dropped: the table of one-pattern slices through the scalar cache (s_load of the entries, once per
// wave): 137 us -- the scalar waits stall the wave in front of its gathers.
//
// Build: k_pat_discover with the wide table; the keys are ranked on the HOST (4096 keys: one small copy each way --
// pattern numbers do not depend on the order the rows arrived in there either); k_pat2_fill copies each pattern from its
// first row and k_pat2_assign compares every row with its pattern entry by entry, bit by bit.
// Chosen when the one-byte table declines for the number of patterns alone, with <= 4096 patterns of <= 64 nonzeros,
// patterns x longest pattern <= 65536 and >= 16 rows per pattern.  LSQRHIP_PAT2=0 never, =1 whenever the limits hold.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PAT2_MAX = 4096;      // patterns (two bytes per row)
constexpr int PAT2_MAX_E = 65536;   // entries of the table: patterns x stride (1 MB)
constexpr int PAT2_TAB = 16384;     // slots of the discovery table
struct __align__(16) PatEnt {
    double v;
    int d;
    int len;   // of the pattern the entry belongs to
};
static_assert(sizeof(PatEnt) == 16, "one 16-byte request per entry");

// len[slot] = length of the first row under the key in that slot (-1: empty slot)
__global__ __launch_bounds__(256) void k_pat2_lens(const int *__restrict__ rowptr,
                                                   const unsigned long long *__restrict__ keys,
                                                   const int *__restrict__ reps, int tab, int *__restrict__ len)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tab) return;
    const int rep = reps[t];
    len[t] = keys[t] != 0ull ? rowptr[rep + 1] - rowptr[rep] : -1;
}

// the entries of every pattern from the first row under its key (slot_pat[slot] = pattern of the slot, -1: empty; the
// table arrives zeroed: an empty pattern is one entry of length 0)
__global__ __launch_bounds__(256) void k_pat2_fill(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                   const double *__restrict__ val, const int *__restrict__ reps,
                                                   const int *__restrict__ slot_pat, int tab, int stride,
                                                   PatEnt *__restrict__ ent)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tab) return;
    const int p = slot_pat[t];
    if (p < 0) return;
    const int rep = reps[t], q0 = rowptr[rep], len = rowptr[rep + 1] - q0;
    for (int k = 0; k < len && k < stride; ++k) {
        PatEnt e;
        e.v = val[q0 + k];
        e.d = col[q0 + k] - rep;
        e.len = len;
        ent[(size_t)p * stride + k] = e;
    }
}

// pid[r] = pattern of row r, after comparing the row with it entry by entry (ctl[1] = 1 on any difference)
__global__ __launch_bounds__(256) void k_pat2_assign(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const double *__restrict__ val, int rows,
                                                     const unsigned long long *__restrict__ keys,
                                                     const int *__restrict__ slot_pat, const PatEnt *__restrict__ ent,
                                                     int tab, int npat, int stride, unsigned short *__restrict__ pid,
                                                     int *__restrict__ ctl)
{
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gstride) {
        if (*(volatile int *)&ctl[1] != 0) return;
        const int q0 = rowptr[r], len = rowptr[r + 1] - q0;
        const unsigned long long h = pat_row_key(col, val, q0, len, (int)r, 1);
        unsigned slot = (unsigned)(h >> 11) & (unsigned)(tab - 1);
        int p = -1;
        for (int tries = 0; tries < tab; ++tries) {
            const unsigned long long k = keys[slot];
            if (k == h) {
                p = slot_pat[slot];
                break;
            }
            if (k == 0ull) break;
            slot = (slot + 1) & (unsigned)(tab - 1);
        }
        bool same = p >= 0 && p < npat && len <= stride && ent[(size_t)p * stride].len == len;
        for (int k = 0; same && k < len; ++k) {
            const PatEnt e = ent[(size_t)p * stride + k];
            same = e.len == len && e.d == col[q0 + k] - (int)r &&
                   __double_as_longlong(e.v) == __double_as_longlong(val[q0 + k]);
        }
        if (!same) {
            ctl[1] = 1;
            return;
        }
        pid[r] = (unsigned short)p;
    }
}

// One slice's entries on their way through the pipeline: q[0] alone when every row of the slice names one pattern
// (`one`, wave-uniform; lane k holds entry k of it), else q[k] = entry k of the lane's own pattern p (-1: no row).
__device__ __forceinline__ uint4 pat2_load(const PatEnt *__restrict__ ent, int e)
{
    return *reinterpret_cast<const uint4 *>(ent + e);
}
__device__ __forceinline__ void pat2_request(int p, bool &one, uint4 (&q)[PAT_K], const PatEnt *__restrict__ ent, int stride,
                                             int lane)
{
    const int p0 = __builtin_amdgcn_readfirstlane(p);   // (rows fill a slice from lane 0: p0 < 0 means no row at all)
    one = __all(p < 0 || p == p0) != 0;
    if (one) {
        q[0] = pat2_load(ent, (p0 >= 0 && lane < stride) ? p0 * stride + lane : 0);
#pragma unroll
        for (int k = 1; k < PAT_K; ++k) q[k] = make_uint4(0u, 0u, 0u, 0u);
    } else {
#pragma unroll
        for (int k = 0; k < PAT_K; ++k) q[k] = pat2_load(ent, (p >= 0 && k < stride) ? p * stride + k : 0);
    }
}
// value, column - row of entry k and the length of the lane's pattern (0: no row)
__device__ __forceinline__ void pat2_unpack(int p, bool one, const uint4 (&q)[PAT_K], double (&v)[PAT_K], int (&d)[PAT_K],
                                            int &len)
{
    if (one) {
#pragma unroll
        for (int k = 0; k < PAT_K; ++k) {   // (k >= stride: entry 0 of the table, never used: k >= len)
            v[k] = __hiloint2double(__builtin_amdgcn_readlane((int)q[0].y, k), __builtin_amdgcn_readlane((int)q[0].x, k));
            d[k] = __builtin_amdgcn_readlane((int)q[0].z, k);
        }
        len = p >= 0 ? __builtin_amdgcn_readlane((int)q[0].w, 0) : 0;
    } else {
#pragma unroll
        for (int k = 0; k < PAT_K; ++k) {
            v[k] = __hiloint2double((int)q[k].y, (int)q[k].x);
            d[k] = (int)q[k].z;
        }
        len = p >= 0 ? (int)q[0].w : 0;
    }
}

template <bool UPD, typename VT = double, bool NT = false, int U = PAT_U>
__global__ __launch_bounds__(SELL_BLOCK, 4) void k_spmv_pat2(
    const unsigned short *__restrict__ pid, const PatEnt *__restrict__ ent, int stride, int rows, int nslices,
    int64_t nblk, const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef,
    const int *__restrict__ stop, double *__restrict__ partials, const double *__restrict__ pin, int npin,
    const NormSlot *__restrict__ slot_in, NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, UpdArgs upd,
    NScale nsc)
{
    __shared__ double red[SELL_BLOCK / WAVE + 1];
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    const int tid = threadIdx.x;
    const bool pre = pin != nullptr && npin <= PAT_SHARE_K * SELL_BLOCK;   // (uniform)
    double pshare[PAT_SHARE_K];
    if (pre) strided_share_load<SELL_BLOCK, PAT_SHARE_K>(pin, npin, pshare);
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdRange xr = xcd_range(nblk, nwg, wg);
    // row of slice u of the trip that starts at block b (-1: none)
    auto row_of = [&](int64_t b, int u) -> int {
        const int64_t bu = b + u * xr.stride;
        const int64_t r64 = (bu * SELL_SLICES + wave) * WAVE + lane;
        return (bu < xr.end && r64 < rows) ? (int)r64 : -1;
    };
    auto pid_of = [&](int64_t b, int (&p)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = row_of(b, u);
            p[u] = r >= 0 ? (int)ld_stream<NT>(&pid[r]) : -1;
        }
    };
    const int64_t step = U * xr.stride;
    int64_t b = xr.first;
    int pid_2[U];   // pattern numbers, two trips ahead of the sums
    pid_of(b, pid_2);

    if (*stop != 0) return;
    SellCoef kc;
    const double share = pre ? strided_share_sum<SELL_BLOCK, PAT_SHARE_K>(pshare, npin) : 0.0;
    if (!sell_prologue<UPD, VT, NT>(coef, pin, npin, slot_in, slot_out, skip_if_zero, upd, nwg, wg, red, kc, nsc, pre, share))
        return;
    const double sx = kc.sx, sy = kc.sy, cy = kc.cy;
    __syncthreads();

    int p_c[U];             // this trip: the lanes' patterns,
    bool one_c[U];          // ... whether a slice names one only,
    uint4 q_c[U][PAT_K];    // ... the entries as requested
#pragma unroll
    for (int u = 0; u < U; ++u) {
        p_c[u] = pid_2[u];
        pat2_request(p_c[u], one_c[u], q_c[u], ent, stride, lane);
    }
    pid_of(b + step, pid_2);
    double sq = 0.0;
    for (; b < xr.end; b += step) {
        int r[U], len[U], d[U][PAT_K];
        double y0[U], v[U][PAT_K], xv[U][PAT_K];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = row_of(b, u);
            pat2_unpack(p_c[u], one_c[u], q_c[u], v[u], d[u], len[u]);
            y0[u] = (double)ld_stream<NT>(&y[r[u] >= 0 ? r[u] : 0]);
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) xv[u][k] = (double)x[k < len[u] ? r[u] + d[u][k] : 0];   // (no entry: x[0], never added)
        }
        // behind the gathers: the next trip's entries, the pattern numbers of the one after it
        int p_n[U];
        bool one_n[U];
        uint4 q_n[U][PAT_K];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            p_n[u] = pid_2[u];
            pat2_request(p_n[u], one_n[u], q_n[u], ent, stride, lane);
        }
        pid_of(b + 2 * step, pid_2);
        double sum[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sum[u] = 0.0;
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                const double p = v[u][k] * (xv[u][k] * sx);
                if (k < len[u]) sum[u] = sum[u] + p;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            for (int k0 = PAT_K; __any(k0 < len[u]); k0 += PAT_K) {   // rows of more than PAT_K nonzeros
                uint4 q2[PAT_K];
                double x2[PAT_K];
#pragma unroll
                for (int k = 0; k < PAT_K; ++k) q2[k] = pat2_load(ent, k0 + k < len[u] ? p_c[u] * stride + k0 + k : 0);
#pragma unroll
                for (int k = 0; k < PAT_K; ++k) x2[k] = (double)x[k0 + k < len[u] ? r[u] + (int)q2[k].z : 0];
#pragma unroll
                for (int k = 0; k < PAT_K; ++k) {
                    const double p = __hiloint2double((int)q2[k].y, (int)q2[k].x) * (x2[k] * sx);
                    if (k0 + k < len[u]) sum[u] = sum[u] + p;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r[u] >= 0) {
                const VT yn = (VT)(cy * (y0[u] * sy) + sum[u]);
                store_through(&y[r[u]], yn);
                const double ys = (double)yn * nsc.s;
                sq += ys * ys;
            }
            p_c[u] = p_n[u];
            one_c[u] = one_n[u];
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) q_c[u][k] = q_n[u][k];
        }
    }
    const double tot = block_sum<SELL_BLOCK>(sq, red);
    if (tid == 0) partials[wg] = tot;
}

// The wide table in PAIRED ROWS (see "PAIRED ROWS" above: lane L owns rows 2L, 2L + 1 of a 128-row group): two pattern
// numbers in one 4-byte load, y read and stored as a pair, and in a group whose 128 rows all name ONE pattern the
// entries in one request and every entry's two gathers of x in one 16-byte gather -- 9 requests per 128 rows where
// k_spmv_pat2 has 20.  A group that crosses an interface requests the entries of both rows of every lane and gathers
// per row.  The same pipeline: pattern numbers of trip t + 2, entries of t + 1, gathered x of t.
// What it buys depends on how many 128-row groups lie inside one region: the 16M-row mesh of 16 x 16 regions (250
// columns each: every second group crosses an interface, every fourth 64-row slice did) 103 -> 101 us; regions of 1000
// columns 86 -> 76 us (profiles/r05/wide_patterns_paired.txt).
template <bool UPD, typename VT = double, bool NT = false>
__global__ __launch_bounds__(SELL_BLOCK, 4) void k_spmv_pat2p(
    const unsigned short *__restrict__ pid, const PatEnt *__restrict__ ent, int stride, int rows, int64_t nblk,
    const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef, const int *__restrict__ stop,
    double *__restrict__ partials, const double *__restrict__ pin, int npin, const NormSlot *__restrict__ slot_in,
    NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, UpdArgs upd, NScale nsc)
{
    typedef typename Vec2<VT>::type V2T;
    __shared__ double red[SELL_BLOCK / WAVE + 1];
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {
        run_rider(rider, red);
        return;
    }
    const int tid = threadIdx.x;
    const bool pre = pin != nullptr && npin <= PAT_SHARE_K * SELL_BLOCK;   // (uniform)
    double pshare[PAT_SHARE_K];
    if (pre) strided_share_load<SELL_BLOCK, PAT_SHARE_K>(pin, npin, pshare);
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdRange xr = xcd_range(nblk, nwg, wg);   // blocks of SELL_SLICES groups of 2 * WAVE rows
    auto row_of = [&](int64_t b) -> int64_t {       // the lane's first row in group (b, wave); >= rows: none
        return b < xr.end ? (b * SELL_SLICES + wave) * (2 * WAVE) + 2 * lane : (int64_t)rows;
    };
    // the pattern numbers of the lane's two rows (pid has two numbers of padding behind the last row)
    auto pids_of = [&](int64_t b) -> unsigned {
        const int64_t r = row_of(b);
        return r < rows ? ld_stream<NT>(reinterpret_cast<const unsigned *>(pid + r)) : 0u;
    };
    // the entries of a group on their way: q0[0] alone when its 128 rows name one pattern (`one`; lane k holds entry k),
    // else q0[k] / q1[k] = entry k of the patterns of the lane's two rows
    auto request = [&](int64_t b, unsigned pp, bool &one, uint4 (&q0)[PAT_K], uint4 (&q1)[PAT_K]) {
        const int p0 = (int)(pp & 0xffffu), p1 = (int)(pp >> 16);
        const int P = __builtin_amdgcn_readfirstlane(p0);
        const bool full = b < xr.end && (b * SELL_SLICES + wave + 1) * (2 * WAVE) <= rows;   // (uniform)
        one = full && __all(p0 == P && p1 == P) != 0;
        if (one) {
            q0[0] = pat2_load(ent, lane < stride ? P * stride + lane : 0);
#pragma unroll
            for (int k = 1; k < PAT_K; ++k) q0[k] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) q1[k] = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                q0[k] = pat2_load(ent, k < stride ? p0 * stride + k : 0);
                q1[k] = pat2_load(ent, k < stride ? p1 * stride + k : 0);
            }
        }
    };
    int64_t b = xr.first;
    unsigned pid_2 = pids_of(b);   // pattern numbers, two trips ahead of the sums

    if (*stop != 0) return;
    SellCoef kc;
    const double share = pre ? strided_share_sum<SELL_BLOCK, PAT_SHARE_K>(pshare, npin) : 0.0;
    if (!sell_prologue<UPD, VT, NT>(coef, pin, npin, slot_in, slot_out, skip_if_zero, upd, nwg, wg, red, kc, nsc, pre, share))
        return;
    const double sx = kc.sx, sy = kc.sy, cy = kc.cy;
    __syncthreads();

    unsigned pp_c = pid_2;
    bool one_c;
    uint4 q0_c[PAT_K], q1_c[PAT_K];
    request(b, pp_c, one_c, q0_c, q1_c);
    pid_2 = pids_of(b + xr.stride);
    double sq = 0.0;
    for (; b < xr.end; b += xr.stride) {
        const int64_t r64 = row_of(b);
        const bool act0 = r64 < rows, act1 = r64 + 1 < rows;
        const int r0 = act0 ? (int)r64 : 0;
        const bool full = (b * SELL_SLICES + wave + 1) * (2 * WAVE) <= rows;   // (uniform)
        double v0[PAT_K], v1[PAT_K], x0[PAT_K], x1[PAT_K], y0, y1;
        int d0[PAT_K], d1[PAT_K], len0, len1;
        if (full) {
            const V2T yv = ld_stream2<NT>(reinterpret_cast<const V2T *>(&y[r0]));
            y0 = (double)yv.x;
            y1 = (double)yv.y;
        } else {
            y0 = act0 ? (double)y[r0] : 0.0;
            y1 = act1 ? (double)y[r0 + 1] : 0.0;
        }
        if (one_c) {
            len0 = len1 = __builtin_amdgcn_readlane((int)q0_c[0].w, 0);
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {   // (k >= stride: entry 0 of the table, never used: k >= len)
                v0[k] = __hiloint2double(__builtin_amdgcn_readlane((int)q0_c[0].y, k), __builtin_amdgcn_readlane((int)q0_c[0].x, k));
                v1[k] = v0[k];
                d0[k] = d1[k] = __builtin_amdgcn_readlane((int)q0_c[0].z, k);
                ld_pair(&x[k < len0 ? r0 + d0[k] : 0], x0[k], x1[k]);   // (no entry: x[0], x[1], never added; the layout needs two columns)
            }
        } else {
            len0 = act0 ? (int)q0_c[0].w : 0;
            len1 = act1 ? (int)q1_c[0].w : 0;
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                v0[k] = __hiloint2double((int)q0_c[k].y, (int)q0_c[k].x);
                v1[k] = __hiloint2double((int)q1_c[k].y, (int)q1_c[k].x);
                d0[k] = (int)q0_c[k].z;
                d1[k] = (int)q1_c[k].z;
                x0[k] = (double)x[k < len0 ? r0 + d0[k] : 0];
                x1[k] = (double)x[k < len1 ? r0 + 1 + d1[k] : 0];
            }
        }
        // behind the gathers: the next trip's entries, the pattern numbers of the one after it
        const unsigned pp_n = pid_2;
        bool one_n;
        uint4 q0_n[PAT_K], q1_n[PAT_K];
        request(b + xr.stride, pp_n, one_n, q0_n, q1_n);
        pid_2 = pids_of(b + 2 * xr.stride);
        double sum0 = 0.0, sum1 = 0.0;
#pragma unroll
        for (int k = 0; k < PAT_K; ++k) {
            const double t0 = v0[k] * (x0[k] * sx), t1 = v1[k] * (x1[k] * sx);
            if (k < len0) sum0 = sum0 + t0;
            if (k < len1) sum1 = sum1 + t1;
        }
        const int p0 = (int)(pp_c & 0xffffu), p1 = (int)(pp_c >> 16);
        for (int k0 = PAT_K; __any(k0 < len0 || k0 < len1); k0 += PAT_K) {   // rows of more than PAT_K nonzeros
            uint4 e0[PAT_K], e1[PAT_K];
            double z0[PAT_K], z1[PAT_K];
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                e0[k] = pat2_load(ent, k0 + k < len0 ? p0 * stride + k0 + k : 0);
                e1[k] = pat2_load(ent, k0 + k < len1 ? p1 * stride + k0 + k : 0);
            }
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                z0[k] = (double)x[k0 + k < len0 ? r0 + (int)e0[k].z : 0];
                z1[k] = (double)x[k0 + k < len1 ? r0 + 1 + (int)e1[k].z : 0];
            }
#pragma unroll
            for (int k = 0; k < PAT_K; ++k) {
                const double t0 = __hiloint2double((int)e0[k].y, (int)e0[k].x) * (z0[k] * sx);
                const double t1 = __hiloint2double((int)e1[k].y, (int)e1[k].x) * (z1[k] * sx);
                if (k0 + k < len0) sum0 = sum0 + t0;
                if (k0 + k < len1) sum1 = sum1 + t1;
            }
        }
        const VT yn0 = (VT)(cy * (y0 * sy) + sum0), yn1 = (VT)(cy * (y1 * sy) + sum1);
        if (full) {
            store_through2(&y[r0], yn0, yn1);
        } else {
            if (act0) store_through(&y[r0], yn0);
            if (act1) store_through(&y[r0 + 1], yn1);
        }
        if (act0) {
            const double ys = (double)yn0 * nsc.s;
            sq += ys * ys;
        }
        if (act1) {
            const double ys = (double)yn1 * nsc.s;
            sq += ys * ys;
        }
        pp_c = pp_n;
        one_c = one_n;
#pragma unroll
        for (int k = 0; k < PAT_K; ++k) {
            q0_c[k] = q0_n[k];
            q1_c[k] = q1_n[k];
        }
    }
    const double tot = block_sum<SELL_BLOCK>(sq, red);
    if (tid == 0) partials[wg] = tot;
}

}  // namespace lsqrhip
