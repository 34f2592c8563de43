// spf_roof.hip -- does a SCALAR-cache prefetch of a read-once stream raise what a CU gets out of HBM?
//
// Every HBM-bound kernel here tops out near 10 bytes per clock and CU: ~64 cache lines in flight in the CU's vector L1
// (TCP) times a ~750-cycle HBM latency.  Scalar loads miss through another path (scalar cache -> L2): if a wave touches,
// with s_load_dword, the lines its vector loads will want a few microseconds later, those vector loads should find them
// in L2 (~220 cycles) and free their TCP slots three times sooner.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/spf_roof.hip -o scripts/_bin/spf_roof && scripts/_bin/spf_roof
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); std::exit(1); } } while (0)

// Each wave streams a contiguous region of `per_wave` bytes in steps of 4 KiB (64 lanes x 4 x 16 B).
// PF = scalar prefetches per step (0 = none), each `stride` bytes apart, `ahead` bytes in front of the vector loads.
template <int PF>
__global__ __launch_bounds__(1024) void k_stream(const uint4 *__restrict__ src, size_t per_wave, int ahead, int stride,
                                                 unsigned *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const char *base = reinterpret_cast<const char *>(src) + wave * per_wave;
    unsigned acc = 0;
    unsigned d[PF > 0 ? PF : 1];
#pragma unroll
    for (int k = 0; k < (PF > 0 ? PF : 1); ++k) d[k] = 0;
    for (size_t off = 0; off < per_wave; off += 4096) {
        if (PF > 0) {
            // the prefetches of the PREVIOUS step have long returned: wait, then let the registers be reused
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the registers stay the loads' until here: an empty asm that "rewrites" each one pins its live range
            // behind the wait -- without it the compiler reads them early and reuses them while loads are in flight)
#pragma unroll
            for (int k = 0; k < PF; ++k) asm volatile("" : "+s"(d[k]));
#pragma unroll
            for (int k = 0; k < PF; ++k) acc += d[k] & 1u;
            size_t p = off + (size_t)ahead;
            if (p + (size_t)PF * stride > per_wave) p = 0;
            const char *q = base + p;
            const unsigned long long qa = (unsigned long long)q;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)qa), hi = __builtin_amdgcn_readfirstlane((unsigned)(qa >> 32));
            const unsigned long long qs = ((unsigned long long)hi << 32) | lo;
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const unsigned long long a = qs + (unsigned long long)k * stride;
                asm volatile("s_load_dword %0, %1, 0x0" : "=s"(d[k]) : "s"(a) : "memory");
            }
        }
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u *v = reinterpret_cast<const v4u *>(base + off) + lane;
        const v4u a0 = __builtin_nontemporal_load(v), a1 = __builtin_nontemporal_load(v + 64),
                  a2 = __builtin_nontemporal_load(v + 128), a3 = __builtin_nontemporal_load(v + 192);
        const v4u t = a0 ^ a1 ^ a2 ^ a3;
        acc += t.x ^ t.y ^ t.z ^ t.w;
    }
    if (PF > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < PF; ++k) asm volatile("" : "+s"(d[k]));
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// scalar loads only: how many bytes per second does the scalar path pull through L2 from HBM?
template <int PF>
__global__ __launch_bounds__(1024) void k_scalar_only(const uint4 *__restrict__ src, size_t per_wave, int stride,
                                                      unsigned *__restrict__ out)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const char *base = reinterpret_cast<const char *>(src) + wave * per_wave;
    unsigned acc = 0;
    unsigned d[PF];
    for (size_t off = 0; off + (size_t)PF * stride <= per_wave; off += (size_t)PF * stride) {
        const unsigned long long qa = (unsigned long long)(base + off);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)qa), hi = __builtin_amdgcn_readfirstlane((unsigned)(qa >> 32));
        const unsigned long long qs = ((unsigned long long)hi << 32) | lo;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const unsigned long long a = qs + (unsigned long long)k * stride;
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(d[k]) : "s"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < PF; ++k) asm volatile("" : "+s"(d[k]));
#pragma unroll
        for (int k = 0; k < PF; ++k) acc += d[k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static double timeit(F &&launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t total = (size_t)8 << 30;     // 8 GiB: far beyond the Infinity Cache
    uint4 *src; unsigned *out;
    CK(hipMalloc(&src, total)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, total));
    std::printf("%-58s %10s %10s %12s\n", "kernel", "ms", "GB/s", "B/clk/CU");
    for (int wpc : {16, 32}) {                // waves per CU: one or two 1024-thread workgroups
        const int grid = 256 * wpc / 16;
        const size_t per_wave = total / ((size_t)grid * 16);
        auto rep = [&](const char *name, double ms) {
            std::fflush(stdout);
            std::printf("%-44s %2d waves/CU %10.3f %10.0f %12.2f\n", name, wpc, ms, total / ms / 1e6, total / (ms * 1e-3) / 2.4e9 / 256);
        };
        std::fprintf(stderr, "wpc %d grid %d per_wave %zu\n", wpc, grid, per_wave);
        rep("vector nt loads only", timeit([&] { hipLaunchKernelGGL(k_stream<0>, dim3(grid), dim3(1024), 0, 0, src, per_wave, 0, 0, out); }, 3));
        for (int ahead : {8192, 16384, 32768, 65536}) {
            char nm[96];
            std::fprintf(stderr, "ahead %d\n", ahead);
            std::snprintf(nm, sizeof nm, "+ 8 s_load / step, 512 B apart, %5d B ahead", ahead);
            rep(nm, timeit([&] { hipLaunchKernelGGL(k_stream<8>, dim3(grid), dim3(1024), 0, 0, src, per_wave, ahead, 512, out); }, 3));
            std::snprintf(nm, sizeof nm, "+ 14 s_load / step, 256 B apart, %5d B ahead", ahead);
            rep(nm, timeit([&] { hipLaunchKernelGGL(k_stream<14>, dim3(grid), dim3(1024), 0, 0, src, per_wave, ahead, 256, out); }, 3));
        }
        std::fprintf(stderr, "scalar only\n");
        rep("scalar loads only, 14 in flight, 128 B apart", timeit([&] { hipLaunchKernelGGL(k_scalar_only<14>, dim3(grid), dim3(1024), 0, 0, src, per_wave, 128, out); }, 2));
        rep("scalar loads only, 14 in flight, 64 B apart", timeit([&] { hipLaunchKernelGGL(k_scalar_only<14>, dim3(grid), dim3(1024), 0, 0, src, per_wave, 64, out); }, 2));
    }
    return 0;
}
