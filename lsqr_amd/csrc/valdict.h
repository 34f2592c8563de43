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
__global__ __launch_bounds__(256) void k_dict_collect(const double *__restrict__ val, int64_t nnz,
                                                      unsigned long long *__restrict__ table, int *__restrict__ ctl)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    volatile int *vctl = ctl;
    unsigned long long prev = VD_EMPTY;  // the previous pattern of this thread is known to be in the table
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride) {
        if (vctl[1] != 0) return;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(val[k]);
        if (bits == prev) continue;
        if (bits == VD_EMPTY) {
            atomicExch(&ctl[1], 1);
            return;
        }
        unsigned h = (unsigned)((bits * 0x9E3779B97F4A7C15ull) >> 52) & (VD_SLOTS - 1);
        for (int probe = 0; probe < VD_SLOTS; ++probe) {
            unsigned long long cur = ((volatile unsigned long long *)table)[h];
            if (cur == VD_EMPTY) {
                cur = atomicCAS(&table[h], VD_EMPTY, bits);
                if (cur == VD_EMPTY) {  // this thread inserted it
                    if (atomicAdd(&ctl[0], 1) + 1 > VD_MAX) {
                        atomicExch(&ctl[1], 1);
                        return;
                    }
                    break;
                }
            }
            if (cur == bits) break;
            h = (h + 1) & (VD_SLOTS - 1);
        }
        prev = bits;
    }
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
