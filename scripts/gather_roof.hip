// gather_roof.hip -- the ceiling for SpMV with scattered columns on one MI355X.
//
// A CSR product with random columns does, per nonzero, one 12-byte coalesced stream read
// (value + column) and one 8-byte gather of x[col].  This microbenchmark does exactly that and
// nothing else (no rows, no reduction tree, no LDS): every lane keeps `U` nonzeros in flight,
// 8 workgroups of 256 threads per CU, persistent grid.  Sweeping the size of x shows where the
// gathers are served from (L2 slice of an XCD, Infinity Cache, HBM) and how many gathers per
// second the memory pipeline sustains -- the roofline the panel / window kernels are held
// against in DESIGN.md 4.2.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_roof.hip -o gpurun_out/gather_roof
//   gpurun_out/gather_roof            # prints a table
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                          \
    do {                                                                               \
        hipError_t _e = (e);                                                           \
        if (_e != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void k_fill(int *col, double *val, int64_t nnz, int ncols, int xcd_local)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        col[i] = (int)(mix((uint64_t)i) % (uint64_t)ncols);
        val[i] = 1.0 + (double)(i & 7);
    }
    (void)xcd_local;
}

template <int U>
__global__ __launch_bounds__(256, 8) void k_gather(const int *__restrict__ col, const double *__restrict__ val,
                                                   const double *__restrict__ x, int64_t nnz, double *__restrict__ out)
{
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256 * U;
    for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < nnz; base += stride) {
        int c[U];
        double a[U], xv[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t i = base + (int64_t)j * 256;
            const int64_t k = i < nnz ? i : nnz - 1;
            c[j] = col[k];
            a[j] = val[k];
        }
#pragma unroll
        for (int j = 0; j < U; ++j) xv[j] = x[c[j]];
#pragma unroll
        for (int j = 0; j < U; ++j) acc += a[j] * xv[j];
    }
    if (acc == 123.456) out[0] = acc;  // keep the work
}

template <int U>
static double run(const int *col, const double *val, const double *x, int64_t nnz, double *out, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<U>, dim3(2048), dim3(256), 0, 0, col, val, x, nnz, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_gather<U>, dim3(2048), dim3(256), 0, 0, col, val, x, nnz, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const int64_t nnz = 400000000ll;  // 4.8 GB of (val, col): far beyond every cache
    int *col;
    double *val, *x, *out;
    CK(hipMalloc(&col, sizeof(int) * nnz));
    CK(hipMalloc(&val, sizeof(double) * nnz));
    CK(hipMalloc(&out, 64));
    const size_t xmax = 1ull << 30;  // 1 GB of x at most
    CK(hipMalloc(&x, xmax));
    CK(hipMemset(x, 0, xmax));
    std::printf("%-14s %-4s %10s %12s %14s %12s\n", "x bytes", "U", "ms", "Ggather/s", "gath/clk/CU", "stream GB/s");
    const size_t sizes[] = {256u << 10, 2u << 20, 16u << 20, 80u << 20, 800u << 20};
    for (size_t xb : sizes) {
        const int ncols = (int)(xb / 8);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, col, val, nnz, ncols, 0);
        CK(hipDeviceSynchronize());
        for (int U : {4, 8}) {
            const double ms = U == 4 ? run<4>(col, val, x, nnz, out, 5) : run<8>(col, val, x, nnz, out, 5);
            const double gps = nnz / (ms * 1e-3) / 1e9;
            std::printf("%-14zu %-4d %10.3f %12.1f %14.3f %12.0f\n", xb, U, ms, gps, gps * 1e9 / (256.0 * 2.4e9),
                        12.0 * nnz / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
