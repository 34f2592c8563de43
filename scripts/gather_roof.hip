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

// Same work, each lane taking 4 CONSECUTIVE nonzeros per step: (val, col) arrive as 16-byte loads
// (two dwordx4 of values, one dwordx4 of columns) instead of 8- and 4-byte ones.  MODE 0: gathers
// from global memory, 1: no gather at all (pure stream), 2: gathers from an LDS copy of x (<= 56 KB).
template <int MODE>
__global__ __launch_bounds__(256, 8) void k_gather_wide(const int *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ x, int64_t nnz, int ncols,
                                                        double *__restrict__ out)
{
    __shared__ double xs[MODE == 2 ? 7168 : 1];
    if (MODE == 2) {
        for (int i = threadIdx.x; i < 7168; i += 256) xs[i] = x[i % ncols];
        __syncthreads();
    }
    double acc = 0.0;
    const int64_t n4 = nnz >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int4 *col4 = reinterpret_cast<const int4 *>(col);
    const double2 *val2 = reinterpret_cast<const double2 *>(val);
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
        const int4 c = col4[q];
        const double2 a01 = val2[2 * q], a23 = val2[2 * q + 1];
        double x0 = 1.0, x1 = 1.0, x2 = 1.0, x3 = 1.0;
        if (MODE == 0) { x0 = x[c.x]; x1 = x[c.y]; x2 = x[c.z]; x3 = x[c.w]; }
        if (MODE == 2) { x0 = xs[c.x % 7168]; x1 = xs[c.y % 7168]; x2 = xs[c.z % 7168]; x3 = xs[c.w % 7168]; }
        if (MODE == 1) { x0 = (double)c.x; x1 = (double)c.y; x2 = (double)c.z; x3 = (double)c.w; }
        acc += a01.x * x0 + a01.y * x1 + a23.x * x2 + a23.y * x3;
    }
    if (acc == 123.456) out[0] = acc;
}

template <int MODE>
static double run_wide(const int *col, const double *val, const double *x, int64_t nnz, int ncols, double *out, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather_wide<MODE>, dim3(2048), dim3(256), 0, 0, col, val, x, nnz, ncols, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(k_gather_wide<MODE>, dim3(2048), dim3(256), 0, 0, col, val, x, nnz, ncols, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static int g_grid = 2048;  // workgroups of 256 threads (8 per CU); argv[1] overrides

template <int U>
static double run(const int *col, const double *val, const double *x, int64_t nnz, double *out, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<U>, dim3(g_grid), dim3(256), 0, 0, col, val, x, nnz, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_gather<U>, dim3(g_grid), dim3(256), 0, 0, col, val, x, nnz, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    if (argc > 1) g_grid = std::atoi(argv[1]);
    std::printf("grid %d workgroups of 256 threads\n", g_grid);
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
    std::printf("\nwide loads (4 consecutive nonzeros per lane, 16-byte loads):\n%-26s %10s %12s %12s\n", "variant", "ms",
                "Gnnz/s", "stream GB/s");
    {
        const int ncols = (2u << 20) / 8;
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, col, val, nnz, ncols, 0);
        CK(hipDeviceSynchronize());
        const double t1 = run_wide<1>(col, val, x, nnz, ncols, out, 5);
        const double t0 = run_wide<0>(col, val, x, nnz, ncols, out, 5);
        const double t2 = run_wide<2>(col, val, x, nnz, ncols, out, 5);
        const double ts[3] = {t1, t0, t2};
        const char *nm[3] = {"stream only", "gather from 2 MB (L2)", "gather from LDS"};
        for (int k = 0; k < 3; ++k)
            std::printf("%-26s %10.3f %12.1f %12.0f\n", nm[k], ts[k], nnz / (ts[k] * 1e-3) / 1e9,
                        12.0 * nnz / (ts[k] * 1e-3) / 1e9);
    }
    return 0;
}
