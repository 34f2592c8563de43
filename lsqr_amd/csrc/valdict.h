// valdict.h -- value dictionary for matrices with few distinct nonzero values.
//
// The reference stores every a(k) as a real64 (src/lsqr.f90:44-46) and so does the CSR here:
// 8 of the 12 (or 10) bytes a nonzero costs on a kernel that is bound by bytes.  Stencil,
// incidence and 0/1 matrices (BASELINE.json configs[1]: the 5-point Laplacian has the two
// values 4 and -1) carry far less information than that.  When the WHOLE matrix has at most
// 256 distinct value bit patterns the build keeps a sorted table of them (dict[256]) and one
// byte per nonzero; the SpMV looks the double up in LDS.  The looked-up double is the stored
// one, bit for bit (-0.0 and NaN payloads are distinct patterns), so every result is
// identical to the 8-byte path (tests/test_gpu_panels.py pins that with LSQRHIP_VAL8=0).
//
//   k_dict_collect   insert every value pattern into a 4096-slot open-addressing table;
//                    gives up (overflow flag, early exit everywhere) past 256 distinct
//   host             sorts the <= 256 patterns (deterministic codes), uploads dict[]
//   k_dict_encode    val[k] -> code[k] by binary search in the sorted table (LDS)
#pragma once

#include "common.h"

namespace lsqrhip {

constexpr int VD_SLOTS = 4096;
constexpr int VD_MAX = 256;
constexpr unsigned long long VD_EMPTY = 0xFFFFFFFFFFFFFFFFull;  // a NaN pattern; a value equal to it disables the dictionary

// table[VD_SLOTS] preset to VD_EMPTY; ctl[0] = distinct count, ctl[1] = overflow flag.
// A per-workgroup LDS filter remembers the patterns this workgroup has already seen in the
// global table, so a matrix with a handful of values does not hammer the same few L2 lines
// with one probe per nonzero (Poisson: 2.7 ms -> ~0.1 ms for 5e6 nonzeros).
__global__ __launch_bounds__(256) void k_dict_collect(const double *__restrict__ val, int64_t nnz,
                                                      unsigned long long *__restrict__ table, int *__restrict__ ctl)
{
    __shared__ unsigned long long seen[512];
    __shared__ int giveup;
    for (int i = threadIdx.x; i < 512; i += 256) seen[i] = VD_EMPTY;
    if (threadIdx.x == 0) giveup = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    volatile int *vctl = ctl;
    volatile int *vgive = &giveup;
    int iter = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride, ++iter) {
        if (*vgive != 0) return;
        if ((iter & 63) == 0 && vctl[1] != 0) {  // another workgroup overflowed
            *vgive = 1;
            return;
        }
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val[k]);
        // splitmix64 finaliser: doubles such as 4.0 or -1.0 have 52 zero low bits, a single
        // multiply would leave the low hash bits zero and every value in the same filter slot
        unsigned long long z = bits;
        z ^= z >> 31;
        z *= 0x9E3779B97F4A7C15ull;
        z ^= z >> 29;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 32;
        const unsigned hs = (unsigned)z;
        volatile unsigned long long *vs = seen;
        if (vs[hs & 511] == bits) continue;  // already in the global table
        if (bits == VD_EMPTY) {
            atomicExch(&ctl[1], 1);
            *vgive = 1;
            return;
        }
        // a value this workgroup has not seen: look at the overflow flag before touching the
        // shared table (once a matrix with arbitrary values has overflowed it, every remaining
        // thread leaves here on its first element instead of queueing up on one atomic counter:
        // 459 ms -> well under 1 ms for 4e8 random values)
        if (vctl[1] != 0) {
            *vgive = 1;
            return;
        }
        unsigned h = (hs >> 9) & (VD_SLOTS - 1);
        for (int probe = 0; probe < VD_SLOTS; ++probe) {
            unsigned long long cur = ((volatile unsigned long long *)table)[h];
            if (cur == VD_EMPTY) {
                cur = atomicCAS(&table[h], VD_EMPTY, bits);
                if (cur == VD_EMPTY) {  // this thread inserted it
                    if (atomicAdd(&ctl[0], 1) + 1 > VD_MAX) {
                        atomicExch(&ctl[1], 1);
                        *vgive = 1;
                        return;
                    }
                    break;
                }
            }
            if (cur == bits) break;
            h = (h + 1) & (VD_SLOTS - 1);
        }
        vs[hs & 511] = bits;  // benign race: a lost update only costs a redundant probe
    }
}

// *missing |= 1 if some val[k] is not in the ascending table dict_bits[nd]: the build only uses
// a dictionary that provably holds every value of the matrix.
__global__ __launch_bounds__(256) void k_dict_verify(const double *__restrict__ val, int64_t nnz,
                                                     const unsigned long long *__restrict__ dict_bits, int nd,
                                                     int *__restrict__ missing)
{
    __shared__ unsigned long long tab[VD_MAX];
    if ((int)threadIdx.x < nd) tab[threadIdx.x] = dict_bits[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int bad = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val[k]);
        int lo = 0, hi = nd - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tab[mid] < bits) lo = mid + 1;
            else hi = mid;
        }
        if (tab[lo] != bits) bad = 1;
    }
    if (bad) atomicOr(missing, 1);
}

// code[k] = index of val[k]'s bit pattern in the ascending table dict_bits[nd] (nd <= 256).
__global__ __launch_bounds__(256) void k_dict_encode(const double *__restrict__ val, int64_t nnz,
                                                     const unsigned long long *__restrict__ dict_bits, int nd,
                                                     unsigned char *__restrict__ code)
{
    __shared__ unsigned long long tab[VD_MAX];
    if ((int)threadIdx.x < nd) tab[threadIdx.x] = dict_bits[threadIdx.x];
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val[k]);
        int lo = 0, hi = nd - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tab[mid] < bits) lo = mid + 1;
            else hi = mid;
        }
        code[k] = (unsigned char)lo;
    }
}

}  // namespace lsqrhip
