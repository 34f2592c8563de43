// csb_roof.hip -- what bounds a product over "column-swept row blocks" on one MI355X.
//
// Question behind it: gather_roof.hip shows that 8-byte gathers of x from an
// XCD's L2 top out at ~180 G/s when every lane of a wave touches its own cache line.  If the
// nonzeros of a row block are processed in COLUMN order, the 64 lanes of one gather instruction
// fall on neighbouring lines (a block of R rows of a matrix with d nonzeros per row holds
// R*d/n nonzeros per column).  Does the gather rate then scale with the lines touched, and what do
// the row sums cost when they are accumulated in LDS with ds_add_f64 (one exact "hi" and one
// exact "lo" part per product, so that the order of the adds cannot change a bit)?
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/csb_roof.hip -o scripts/_bin/csb_roof
//   scripts/_bin/csb_roof
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(e)                                                                              \
    do {                                                                                   \
        hipError_t _e = (e);                                                               \
        if (_e != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Block b holds `per` nonzeros.  sorted: column of element j = j * n / per + jitter (ascending,
// uniform); else uniformly random.
__global__ void k_fill(int *col, double *val, int64_t per, int nblocks, int ncols, int sorted)
{
    const int64_t nnz = per * nblocks;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double gap = (double)ncols / (double)per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const int64_t j = i % per;
        const uint64_t h = mix((uint64_t)i);
        int c;
        if (sorted) {
            const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0);
            c = (int)(((double)j + u) * gap);
            if (c >= ncols) c = ncols - 1;
        } else {
            c = (int)(h % (uint64_t)ncols);
        }
        col[i] = c;
        val[i] = 0.25 + (double)(h & 1023) * (1.0 / 1024.0);
    }
}

constexpr int ROWS = 8192;

// One 1024-thread workgroup per CU sweeps whole blocks; wave w takes chunks w, w+16, ... of 64*U
// consecutive nonzeros.  ATOM: 0 = registers only, 1 = one ds_add_f64 per nonzero, 2 = hi + lo.
template <int U, int ATOM, int THREADS>
__global__ __launch_bounds__(THREADS) void k_csb(const int *__restrict__ col, const double *__restrict__ val,
                                                 const double *__restrict__ x, int64_t per, int nblocks, double sx,
                                                 double C0, double C1, double *__restrict__ out)
{
    __shared__ double acc[ATOM == 0 ? 1 : (ATOM == 1 ? ROWS : 2 * ROWS)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = THREADS / 64;
    double keep = 0.0;
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
        if (ATOM) {
            for (int i = tid; i < (ATOM == 1 ? ROWS : 2 * ROWS); i += THREADS) acc[i] = 0.0;
            __syncthreads();
        }
        const int64_t base = (int64_t)b * per;
        const int64_t last = per - 1;
        for (int64_t k0 = (int64_t)wave * 64 * U; k0 < per; k0 += (int64_t)NW * 64 * U) {
            int c[U];
            double a[U], xv[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                int64_t k = k0 + j * 64 + lane;
                k = k < last ? k : last;
                c[j] = col[base + k];
                a[j] = val[base + k];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) xv[j] = x[c[j]];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const double p = a[j] * (xv[j] * sx);
                if (ATOM == 0) {
                    keep += p;
                } else {
                    const int r = (int)(((unsigned)c[j] * 2654435761u) >> 19);  // 13 bits
                    if (ATOM == 1) {
                        atomicAdd(&acc[r], p);
                    } else {
                        const double hi = (p + C0) - C0;
                        const double rem = p - hi;
                        const double lo = (rem + C1) - C1;
                        atomicAdd(&acc[r], hi);
                        atomicAdd(&acc[ROWS + r], lo);
                    }
                }
            }
        }
        if (ATOM) {
            __syncthreads();
            for (int i = tid; i < ROWS; i += THREADS) out[(int64_t)b * ROWS + i] = ATOM == 1 ? acc[i] : acc[i] + acc[ROWS + i];
            __syncthreads();
        }
    }
    if (ATOM == 0 && keep == 123.456) out[0] = keep;
}

template <int U, int ATOM, int THREADS>
static double run(const int *col, const double *val, const double *x, int64_t per, int nblocks, double *out, int grid,
                  int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double C0 = 1.5 * 4503599627370496.0 * 0x1p-36 * 4.0;   // quantum 2^-34: |p| <= 2
    const double C1 = C0 * 0x1p-37;
    hipLaunchKernelGGL((k_csb<U, ATOM, THREADS>), dim3(grid), dim3(THREADS), 0, 0, col, val, x, per, nblocks, 1.0, C0, C1, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_csb<U, ATOM, THREADS>), dim3(grid), dim3(THREADS), 0, 0, col, val, x, per, nblocks, 1.0, C0, C1, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int nblocks = argc > 1 ? std::atoi(argv[1]) : 512;
    const int64_t per_max = 976640;
    const int64_t cap = per_max * nblocks;
    int *col;
    double *val, *x, *out;
    CK(hipMalloc(&col, sizeof(int) * cap));
    CK(hipMalloc(&val, sizeof(double) * cap));
    CK(hipMalloc(&out, sizeof(double) * (size_t)ROWS * nblocks));
    CK(hipMalloc(&x, sizeof(double) * 10000000));
    CK(hipMemset(x, 0, sizeof(double) * 10000000));
    struct Cfg { const char *name; int ncols; int64_t per; int sorted; };
    const Cfg cfgs[] = {
        {"n=1e7 R*d=819200 (config 4: 8192 rows x 100) sorted", 10000000, 819200, 1},
        {"n=1e7 R*d=976640 (config 4: 9766 rows x 100) sorted", 10000000, 976640, 1},
        {"n=1e7 R*d=819200 random order", 10000000, 819200, 0},
        {"n=1e7 R*d=488320 (N=8 shard: 4883 rows x 100) sorted", 10000000, 488320, 1},
        {"n=1e7 R*d=102400 (N=8 shard A': 8192 rows x 12.5, x=1.25e6)", 1250000, 102400, 1},
        {"n=2e6 R*d=204800 (gap 10) sorted", 2000000, 204800, 1},
        {"n=1e6 R*d=819200 (config 3 at 100: gap 1.2) sorted", 1000000, 819200, 1},
        {"n=1e6 R*d=819200 random order", 1000000, 819200, 0},
    };
    std::printf("%-62s %-22s %8s %10s %10s\n", "stream", "kernel", "ms", "Gnnz/s", "GB/s(12B)");
    for (const Cfg &c : cfgs) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, col, val, c.per, nblocks, c.ncols, c.sorted);
        CK(hipDeviceSynchronize());
        const double nnz = (double)c.per * nblocks;
        auto rep = [&](const char *kn, double ms) {
            std::printf("%-62s %-22s %8.3f %10.1f %10.0f\n", c.name, kn, ms, nnz / (ms * 1e-3) / 1e9,
                        12.0 * nnz / (ms * 1e-3) / 1e9);
        };
        rep("1024thr U4 regs", run<4, 0, 1024>(col, val, x, c.per, nblocks, out, 256, 3));
        rep("1024thr U8 regs", run<8, 0, 1024>(col, val, x, c.per, nblocks, out, 256, 3));
        rep("256thr x4/CU U4 regs", run<4, 0, 256>(col, val, x, c.per, nblocks, out, 1024, 3));
        rep("1024thr U4 1 atomic", run<4, 1, 1024>(col, val, x, c.per, nblocks, out, 256, 3));
        rep("1024thr U4 hi+lo", run<4, 2, 1024>(col, val, x, c.per, nblocks, out, 256, 3));
        rep("1024thr U8 hi+lo", run<8, 2, 1024>(col, val, x, c.per, nblocks, out, 256, 3));
    }
    return 0;
}
