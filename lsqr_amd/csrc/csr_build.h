// csr_build.h -- K0: COO -> CSR(A) and COO -> CSR(A') on the device.
//
// Replaces the deep copies of `initialize_ez` (reference src/lsqr.f90:113-118): the
// reference keeps the COO triplets and walks them with scattered read-modify-writes
// on every aprod call; here they are sorted ONCE into two row-major copies so both
// products stream memory and need no atomics.
//
// The sort is a stable LSD radix sort (8-bit digits) of 64-bit words
// (key << 32 | original position): histogram + scan + stable scatter per pass.
// Stability means every CSR row keeps its entries in COO order, duplicates
// included -- the same left-to-right sums as src/lsqr.f90:168-172 / :188-192.
// Already row-sorted input (the common case) skips the sort.
#pragma once

#include "common.h"

namespace lsqrhip {

constexpr int RS_BLOCK = 256;
constexpr int RS_WAVES = RS_BLOCK / WAVE;
constexpr int RS_STEPS = 16;                         // 64-lane steps per wave
constexpr int RS_WAVE_ITEMS = RS_STEPS * WAVE;       // 1024
constexpr int RS_TILE = RS_WAVES * RS_WAVE_ITEMS;    // 4096 elements per workgroup

// packed[i] = (key-1) << 32 | i ; flags[0] |= 1 if a key is outside [1, limit],
// flags[1] |= 1 if keys are not non-decreasing.
__global__ __launch_bounds__(256) void k_pack_keys(const int *__restrict__ keys, int64_t nnz, int limit,
                                                   unsigned long long *__restrict__ packed,
                                                   int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int bad = 0, unsorted = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const int k = keys[i];
        if (k < 1 || k > limit) bad = 1;
        if (i > 0 && keys[i - 1] > k) unsorted = 1;
        packed[i] = ((unsigned long long)(unsigned)(k - 1) << 32) | (unsigned long long)(unsigned)i;
    }
    if (bad) atomicOr(&flags[0], 1);
    if (unsorted) atomicOr(&flags[1], 1);
}

// Column-panel layout (spmv.h "panels"): the sort key is the VIRTUAL row
// panel(col) * rows + (row - 1), so the CSR that comes out is that of the panel-stacked matrix
// [A_panel0; A_panel1; ...] with every virtual row still in COO order.
// flags[0] |= 1 for a bad row index, flags[2] |= 1 for a bad column index.
__global__ __launch_bounds__(256) void k_pack_keys_panel(const int *__restrict__ rowk,
                                                         const int *__restrict__ colk, int64_t nnz, int rows,
                                                         int cols, int pw,
                                                         unsigned long long *__restrict__ packed,
                                                         int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int badr = 0, badc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        int r = rowk[i], c = colk[i];
        if (r < 1 || r > rows) { badr = 1; r = 1; }
        if (c < 1 || c > cols) { badc = 1; c = 1; }
        const unsigned long long key = (unsigned long long)((c - 1) / pw) * (unsigned long long)rows +
                                       (unsigned long long)(r - 1);
        packed[i] = (key << 32) | (unsigned long long)(unsigned)i;
    }
    if (badr) atomicOr(&flags[0], 1);
    if (badc) atomicOr(&flags[2], 1);
}

// How far the entries sit from the (scaled) diagonal: sum |col - row * cols / rows| in integer
// arithmetic (order-independent, so deterministic).  Banded / local matrices score small and
// keep the plain CSR; random column patterns score ~cols/3 and get column panels.
__global__ __launch_bounds__(256) void k_col_deviation(const int *__restrict__ rowk,
                                                       const int *__restrict__ colk, int64_t nnz, int rows,
                                                       int cols, unsigned long long *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const long long r = rowk[i] - 1, c = colk[i] - 1;
        const long long d = c - (r * (long long)cols) / (rows > 0 ? rows : 1);
        acc += (unsigned long long)(d < 0 ? -d : d);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && acc) atomicAdd(out, acc);
}

__global__ __launch_bounds__(RS_BLOCK) void k_radix_hist(const unsigned long long *__restrict__ in,
                                                         int64_t nnz, int shift, int64_t nblocks,
                                                         unsigned *__restrict__ hist_g)
{
    __shared__ unsigned hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    for (int j = 0; j < RS_TILE / RS_BLOCK; ++j) {
        const int64_t i = base + j * RS_BLOCK + threadIdx.x;
        if (i < nnz) atomicAdd(&hist[(unsigned)(in[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist_g[(int64_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// In-place exclusive scan of a[0..L) by ONE workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_exclusive_scan(unsigned *__restrict__ a, int64_t L)
{
    __shared__ unsigned sums[1024];
    const int t = threadIdx.x;
    const int64_t per = (L + 1023) / 1024;
    const int64_t lo = (int64_t)t * per, hi = (lo + per < L) ? lo + per : L;
    unsigned s = 0;
    for (int64_t i = lo; i < hi; ++i) s += a[i];
    sums[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 segment sums
    for (int off = 1; off < 1024; off <<= 1) {
        unsigned v = (t >= off) ? sums[t - off] : 0u;
        __syncthreads();
        sums[t] += v;
        __syncthreads();
    }
    unsigned run = (t == 0) ? 0u : sums[t - 1];
    for (int64_t i = lo; i < hi; ++i) {
        const unsigned v = a[i];
        a[i] = run;
        run += v;
    }
}

// Large scans in three coalesced steps (the one-workgroup kernel above walks L/1024 consecutive
// elements per thread: 68 ms for the 2.5e7 histogram entries of a 4e8-nonzero sort):
//   k_scan_sums   sums[b] = sum of chunk b (SCAN_CHUNK elements)
//   k_exclusive_scan on sums (a few thousand entries)
//   k_scan_apply  in-place exclusive scan of every chunk, offset by sums[b]
constexpr int SCAN_CHUNK = 8192;  // 256 threads x 32

__global__ __launch_bounds__(256) void k_scan_sums(const unsigned *__restrict__ a, int64_t L,
                                                   unsigned *__restrict__ sums)
{
    __shared__ unsigned red[256 / WAVE];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    unsigned s = 0;
    for (int j = 0; j < SCAN_CHUNK / 256; ++j) {
        const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
        if (i < L) s += a[i];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_scan_apply(unsigned *__restrict__ a, int64_t L,
                                                    const unsigned *__restrict__ sums)
{
    __shared__ unsigned wsum[256 / WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    unsigned carry = sums[blockIdx.x];
    for (int j = 0; j < SCAN_CHUNK / 256; ++j) {
        const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
        const unsigned v = i < L ? a[i] : 0u;
        unsigned inc = v;  // inclusive scan within the wave
        for (int off = 1; off < WAVE; off <<= 1) {
            const unsigned t = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += t;
        }
        if (lane == WAVE - 1) wsum[wid] = inc;
        __syncthreads();
        unsigned before = 0, total = 0;
        for (int k = 0; k < 256 / WAVE; ++k) {
            if (k < wid) before += wsum[k];
            total += wsum[k];
        }
        if (i < L) a[i] = carry + before + (inc - v);
        carry += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(RS_BLOCK) void k_radix_scatter(const unsigned long long *__restrict__ in,
                                                            unsigned long long *__restrict__ out,
                                                            int64_t nnz, int shift, int64_t nblocks,
                                                            const unsigned *__restrict__ hist_scanned)
{
    __shared__ unsigned cnt[RS_WAVES][256];
    __shared__ unsigned base[RS_WAVES][256];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
    for (int j = 0; j < RS_WAVES; ++j) cnt[j][threadIdx.x] = 0;
    __syncthreads();
    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * RS_WAVE_ITEMS;
    // A: per-wave digit counts of its contiguous sub-tile
    for (int s = 0; s < RS_STEPS; ++s) {
        const int64_t i = wbase + s * WAVE + lane;
        if (i < nnz) atomicAdd(&cnt[w][(unsigned)(in[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    // B: first output slot of (this workgroup, wave, digit)
    {
        const int d = threadIdx.x;
        unsigned run = hist_scanned[(int64_t)d * nblocks + blockIdx.x];
        for (int j = 0; j < RS_WAVES; ++j) {
            base[j][d] = run;
            run += cnt[j][d];
        }
    }
    __syncthreads();
    // C: each wave walks its sub-tile in order; lanes with the same digit are ranked
    // by lane index (ballot match), so equal keys keep their input order.
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int s = 0; s < RS_STEPS; ++s) {
        const int64_t i = wbase + s * WAVE + lane;
        const bool valid = i < nnz;
        const unsigned long long p = valid ? in[i] : 0ull;
        const unsigned d = (unsigned)(p >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool one = (d >> bit) & 1u;
            const unsigned long long bal = __ballot(valid && one);
            peers &= one ? bal : ~bal;
        }
        const unsigned rank = (unsigned)__popcll(peers & lt);
        const unsigned count = (unsigned)__popcll(peers);
        unsigned slot = 0;
        if (valid) slot = base[w][d];
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            out[(int64_t)slot + rank] = p;
            if (rank == 0) base[w][d] = slot + count;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// out_col[i] = other[idx]-1, out_val[i] = a[idx]  with idx = low word of sorted[i]
__global__ __launch_bounds__(256) void k_csr_gather(const unsigned long long *__restrict__ sorted,
                                                    int64_t nnz, const int *__restrict__ other,
                                                    const double *__restrict__ a,
                                                    int *__restrict__ out_col,
                                                    double *__restrict__ out_val)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const unsigned idx = (unsigned)(sorted[i] & 0xffffffffull);
        out_col[i] = other[idx] - 1;
        out_val[i] = a[idx];
    }
}

// rowptr[r] = first position whose key is >= r   (keys sorted; rowptr[rows] = nnz)
template <typename OffT>
__global__ __launch_bounds__(256) void k_rowptr_from_sorted(const unsigned long long *__restrict__ sorted,
                                                            int64_t nnz, int rows,
                                                            OffT *__restrict__ rowptr)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nnz; i += stride) {
        const int64_t key = (i < nnz) ? (int64_t)(sorted[i] >> 32) : (int64_t)rows;
        const int64_t prev = (i > 0) ? (int64_t)(sorted[i - 1] >> 32) : -1;
        for (int64_t r = prev + 1; r <= key; ++r) rowptr[r] = (OffT)i;
    }
}

}  // namespace lsqrhip
