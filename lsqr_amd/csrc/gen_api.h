// gen_api.h -- synthetic sparse systems generated directly in HBM (included by lsqrhip.hip).
//
// The benchmark configurations of BASELINE.json reach 10^9 nonzeros; generating them on the
// host and pushing 16 B/nnz through PCIe would dominate every run.  These kernels emit the
// same (irow, icol, a, b) as lsqr_amd/problems.py, bit for bit: both sides evaluate one
// counter-based hash (splitmix64 of (seed, stream, row, t)) in integer arithmetic.
#pragma once

namespace lsqrhip {

__device__ __forceinline__ unsigned long long sm64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long rng_u64(unsigned long long seed, unsigned stream,
                                                      unsigned long long i, unsigned long long t)
{
    unsigned long long h = sm64(seed ^ ((unsigned long long)stream * 0x632BE59BD9B4E019ull));
    h = sm64(h ^ i);
    return sm64(h ^ t);
}
__device__ __forceinline__ int u64_to_index(unsigned long long h, long long n)
{
    return (int)(((h >> 32) * (unsigned long long)n) >> 32);
}
__device__ __forceinline__ double u64_to_unit(unsigned long long h)
{
    return (double)(h >> 11) * 0x1.0p-52 - 1.0;
}
constexpr unsigned S_COL = 1, S_VAL = 2, S_B = 3;

// kind 0: every row holds `per_row` draws (duplicates kept)
__global__ __launch_bounds__(256) void k_gen_random(unsigned long long seed, long long n, long long per_row,
                                                    long long row0, long long nrows, int *__restrict__ irow,
                                                    int *__restrict__ icol, double *__restrict__ a,
                                                    double *__restrict__ b)
{
    const long long total = nrows * per_row;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long r = e / per_row, t = e - r * per_row;
        const unsigned long long k = (unsigned long long)(row0 + r);
        irow[e] = (int)(r + 1);
        icol[e] = u64_to_index(rng_u64(seed, S_COL, k, (unsigned long long)t), n) + 1;
        a[e] = u64_to_unit(rng_u64(seed, S_VAL, k, (unsigned long long)t));
        if (t == 0 && b) b[r] = u64_to_unit(rng_u64(seed, S_B, k, 0ull));
    }
}

// kind 2: row degrees given by a local row pointer (host-built from the integer CDF table)
__global__ __launch_bounds__(256) void k_gen_by_rowptr(unsigned long long seed, long long n, long long row0,
                                                       long long nrows, const long long *__restrict__ rowptr,
                                                       int *__restrict__ irow, int *__restrict__ icol,
                                                       double *__restrict__ a, double *__restrict__ b)
{
    const long long total = rowptr[nrows];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        long long lo = 0, hi = nrows;  // last r with rowptr[r] <= e
        while (hi - lo > 1) {
            const long long mid = (lo + hi) >> 1;
            if (rowptr[mid] <= e) lo = mid;
            else hi = mid;
        }
        const long long r = lo, t = e - rowptr[r];
        const unsigned long long k = (unsigned long long)(row0 + r);
        irow[e] = (int)(r + 1);
        icol[e] = u64_to_index(rng_u64(seed, S_COL, k, (unsigned long long)t), n) + 1;
        a[e] = u64_to_unit(rng_u64(seed, S_VAL, k, (unsigned long long)t));
    }
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows && b; r += stride)
        b[r] = u64_to_unit(rng_u64(seed, S_B, (unsigned long long)(row0 + r), 0ull));
}

// 5-point Poisson: number of entries in rows [0, k)
__host__ __device__ inline long long poisson_offset(long long k, long long nx, long long ny)
{
    const long long top = k < nx ? k : nx;                       // rows with j == 0
    const long long left = (k + nx - 1) / nx;                    // rows with i == 0
    const long long right = k / nx;                              // rows with i == nx-1
    const long long bot = k - nx * (ny - 1) > 0 ? k - nx * (ny - 1) : 0;  // rows with j == ny-1
    return 5 * k - top - left - right - bot;
}

// kind 1: row k = j*nx + i: -1 at (i,j-1), (i-1,j), 4, -1 at (i+1,j), (i,j+1), ascending columns
__global__ __launch_bounds__(256) void k_gen_poisson(long long nx, long long ny, long long row0, long long nrows,
                                                     int *__restrict__ irow, int *__restrict__ icol,
                                                     double *__restrict__ a)
{
    const long long base = poisson_offset(row0, nx, ny);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += stride) {
        const long long k = row0 + r, i = k % nx, j = k / nx;
        long long p = poisson_offset(k, nx, ny) - base;
        const int lr = (int)(r + 1);
        if (j > 0) { irow[p] = lr; icol[p] = (int)(k - nx + 1); a[p] = -1.0; ++p; }
        if (i > 0) { irow[p] = lr; icol[p] = (int)(k); a[p] = -1.0; ++p; }
        irow[p] = lr; icol[p] = (int)(k + 1); a[p] = 4.0; ++p;
        if (i < nx - 1) { irow[p] = lr; icol[p] = (int)(k + 2); a[p] = -1.0; ++p; }
        if (j < ny - 1) { irow[p] = lr; icol[p] = (int)(k + nx + 1); a[p] = -1.0; ++p; }
    }
}

// kind 3: the five-point mesh of problems.mesh2d -- k of a cell = 2 + u(-1, 1) of its region, a face the harmonic
// mean of its two cells (the cell's own k on the boundary of the grid, kept in the diagonal only); the structure is
// Poisson's.  Every operation in the order the host generator has it (no contraction: -ffp-contract=off).
__global__ __launch_bounds__(256) void k_gen_mesh(unsigned long long seed, long long nx, long long ny, long long bx,
                                                  long long by, long long row0, long long nrows, int *__restrict__ irow,
                                                  int *__restrict__ icol, double *__restrict__ a, double *__restrict__ b)
{
    const long long base = poisson_offset(row0, nx, ny);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += stride) {
        const long long c = row0 + r, i = c % nx, j = c / nx;
        auto kcell = [&](long long ii, long long jj) {
            const unsigned long long region = (unsigned long long)((ii * bx) / nx + bx * ((jj * by) / ny));
            return 2.0 + u64_to_unit(rng_u64(seed, S_VAL, region, 0ull));
        };
        const double k = kcell(i, j);
        auto face = [&](bool ok, long long ii, long long jj) {
            if (!ok) return k;
            const double kn = kcell(ii, jj);
            return 2.0 * k * kn / (k + kn);
        };
        const double fs = face(j > 0, i, j - 1), fw = face(i > 0, i - 1, j), fe = face(i < nx - 1, i + 1, j),
                     fn = face(j < ny - 1, i, j + 1);
        long long p = poisson_offset(c, nx, ny) - base;
        const int lr = (int)(r + 1);
        if (j > 0) { irow[p] = lr; icol[p] = (int)(c - nx + 1); a[p] = -fs; ++p; }
        if (i > 0) { irow[p] = lr; icol[p] = (int)(c); a[p] = -fw; ++p; }
        irow[p] = lr; icol[p] = (int)(c + 1); a[p] = ((fs + fw) + fe) + fn; ++p;
        if (i < nx - 1) { irow[p] = lr; icol[p] = (int)(c + 2); a[p] = -fe; ++p; }
        if (j < ny - 1) { irow[p] = lr; icol[p] = (int)(c + nx + 1); a[p] = -fn; ++p; }
        if (b) b[r] = u64_to_unit(rng_u64(seed, S_B, (unsigned long long)c, 0ull));
    }
}

}  // namespace lsqrhip

extern "C" int64_t lsqrhip_gen_count(int kind, int64_t m, int64_t n, int64_t p0, int64_t p1, int64_t row0,
                                     int64_t nrows)
{
    (void)m;
    (void)n;
    if (kind == 0) return nrows * p0;
    if (kind == 1) return poisson_offset(row0 + nrows, p0, p1) - poisson_offset(row0, p0, p1);
    if (kind == 3 && p0 > 0) return poisson_offset(row0 + nrows, p0, m / p0) - poisson_offset(row0, p0, m / p0);
    return -1;
}

extern "C" int lsqrhip_gen_coo(int kind, uint64_t seed, int64_t m, int64_t n, int64_t p0, int64_t p1, int64_t row0,
                               int64_t nrows, const int64_t *d_rowptr_local, int *d_irow, int *d_icol, double *d_a,
                               double *d_b, int64_t *nnz_out)
{
    if (!d_irow || !d_icol || !d_a || nrows < 0 || row0 < 0 || row0 + nrows > m)
        return fail(LSQRHIP_ERR_ARG, "bad generator arguments");
    if (nrows >= (1ll << 31) || n >= (1ll << 31)) return fail(LSQRHIP_ERR_TOO_LARGE, "rows/cols must fit int32");
    RET(use_device());
    int64_t nnz = 0;
    const unsigned grid = 4096;
    if (kind == 0) {
        nnz = nrows * p0;
        hipLaunchKernelGGL(k_gen_random, dim3(grid), dim3(256), 0, 0, (unsigned long long)seed, (long long)n,
                           (long long)p0, (long long)row0, (long long)nrows, d_irow, d_icol, d_a, d_b);
    } else if (kind == 1) {
        if (p0 * p1 != m || n != m) return fail(LSQRHIP_ERR_ARG, "poisson: m = n = nx*ny required");
        nnz = poisson_offset(row0 + nrows, p0, p1) - poisson_offset(row0, p0, p1);
        hipLaunchKernelGGL(k_gen_poisson, dim3(grid), dim3(256), 0, 0, (long long)p0, (long long)p1, (long long)row0,
                           (long long)nrows, d_irow, d_icol, d_a);
    } else if (kind == 3) {   // p0 = nx (ny = m / nx), p1 = bx << 16 | by
        const int64_t nx = p0, ny = nx > 0 ? m / nx : 0, bx = p1 >> 16, by = p1 & 0xffff;
        if (nx <= 0 || nx * ny != m || n != m || bx <= 0 || by <= 0 || bx > nx || by > ny)
            return fail(LSQRHIP_ERR_ARG, "mesh: m = n = nx*ny and 1 <= bx <= nx, 1 <= by <= ny required");
        nnz = poisson_offset(row0 + nrows, nx, ny) - poisson_offset(row0, nx, ny);
        hipLaunchKernelGGL(k_gen_mesh, dim3(grid), dim3(256), 0, 0, (unsigned long long)seed, (long long)nx, (long long)ny,
                           (long long)bx, (long long)by, (long long)row0, (long long)nrows, d_irow, d_icol, d_a, d_b);
    } else if (kind == 2) {
        if (!d_rowptr_local) return fail(LSQRHIP_ERR_ARG, "power-law generator needs the local row pointer");
        long long last = 0;
        HIPCHK(hipMemcpy(&last, d_rowptr_local + nrows, sizeof(last), hipMemcpyDeviceToHost));
        nnz = last;
        hipLaunchKernelGGL(k_gen_by_rowptr, dim3(grid), dim3(256), 0, 0, (unsigned long long)seed, (long long)n,
                           (long long)row0, (long long)nrows, (const long long *)d_rowptr_local, d_irow, d_icol,
                           d_a, d_b);
    } else {
        return fail(LSQRHIP_ERR_ARG, "unknown generator kind");
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if (nnz_out) *nnz_out = nnz;
    return LSQRHIP_OK;
}
