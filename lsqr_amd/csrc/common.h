// common.h -- shared device helpers for the LSQR HIP kernels (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsqrhip {

constexpr int WAVE = 64;

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

// Blocks b and b+8 share an XCD (and its 4 MiB L2) under the observed round-robin
// placement.  Give each XCD label one contiguous eighth of the work items so that
// neighbouring row blocks -- which gather neighbouring parts of x -- share an L2.
// Placement only changes speed, never results.
struct XcdRange {
    int64_t first, end, stride;
};
__device__ __forceinline__ XcdRange xcd_range(int64_t nitems)
{
    const int g = gridDim.x;
    XcdRange r;
    if ((g & 7) != 0 || g < 8) {
        r.first = blockIdx.x;
        r.end = nitems;
        r.stride = g;
        return r;
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int64_t per = (nitems + 7) >> 3;
    r.first = (int64_t)xcd * per + slot;
    r.end = (int64_t)(xcd + 1) * per;
    if (r.end > nitems) r.end = nitems;
    r.stride = g >> 3;
    return r;
}

}  // namespace lsqrhip
