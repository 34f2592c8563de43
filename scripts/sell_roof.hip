// sell_roof.hip -- what limits the sliced-ELL product at config 2 (5-point Poisson, 1-byte value
// codes, 16-bit slice-relative columns), and what would a packed-record layout buy?
//
// The product moves 40 B per row (15 B of matrix + 1 B row length + x + y read and written); a
// pure streaming kernel moves 40 MB in ~5 us on this chip (scripts/grid_barrier.hip, data in the
// Infinity Cache) while k_spmv_sell takes ~10 us.  Variants, all computing the same
// y <- cy*(y*sy) + A (x*sx) and the same per-workgroup sum of squares, no lazy-norm prologue:
//   A  the layout of csrc/sell.h: per k one 2-byte column load and one 1-byte code load per lane,
//      descriptors (slice offset, column base) fetched with vector loads
//   B  A with the wave index made scalar: descriptors arrive through the scalar cache
//   C  packed records: one 16-byte load per row = 5 columns (u16) + 5 codes (u8) + length (u8)
//   D  C with two slices per wave in flight
//   S  stream only: 16 B record + y + x[row] (no gather), y written
// on nx x ny = 1000^2 and 2000^2, persistent grid of 2048 workgroups (XCD-contiguous) or one
// workgroup per 4 slices.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/sell_roof.hip -o scripts/_bin/sell_roof
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(e)                                                                          \
    do {                                                                               \
        hipError_t _e = (e);                                                           \
        if (_e != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

constexpr int W = 5;

struct Mat {
    int rows, nslices;
    unsigned *soff;         // nslices + 1
    int *cbase;             // nslices
    unsigned short *sc16;   // 64 W per slice, column-major
    unsigned char *sv8, *rlen;
    uint4 *rec;             // rows (padded to slices)
    double *dict;           // 256
};

__global__ void k_build(Mat m, int nx, int ny)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = r >> 6, lane = r & 63;
    if (s >= m.nslices) return;
    const int r0 = s * 64;
    const int cb = max(0, r0 - nx);
    if (lane == 0) {
        m.soff[s] = (unsigned)s * 64u * W;
        if (s == m.nslices - 1) m.soff[s + 1] = (unsigned)(s + 1) * 64u * W;
        m.cbase[s] = cb;
    }
    int c[W], code[W], len = 0;
    if (r < m.rows) {
        const int i = r % nx, j = r / nx;
        if (j > 0) { c[len] = r - nx; code[len++] = 1; }
        if (i > 0) { c[len] = r - 1; code[len++] = 1; }
        c[len] = r; code[len++] = 0;
        if (i < nx - 1) { c[len] = r + 1; code[len++] = 1; }
        if (j < ny - 1) { c[len] = r + nx; code[len++] = 1; }
    }
    for (int k = len; k < W; ++k) { c[k] = cb; code[k] = 0; }
    for (int k = 0; k < W; ++k) {
        m.sc16[(size_t)s * 64 * W + 64 * k + lane] = (unsigned short)(c[k] - cb);
        m.sv8[(size_t)s * 64 * W + 64 * k + lane] = (unsigned char)code[k];
    }
    if (r < m.rows) m.rlen[r] = (unsigned char)len;
    uint4 q;
    q.x = (unsigned)(c[0] - cb) | ((unsigned)(c[1] - cb) << 16);
    q.y = (unsigned)(c[2] - cb) | ((unsigned)(c[3] - cb) << 16);
    q.z = (unsigned)(c[4] - cb) | ((unsigned)code[0] << 16) | ((unsigned)code[1] << 24);
    q.w = (unsigned)code[2] | ((unsigned)code[3] << 8) | ((unsigned)code[4] << 16) | ((unsigned)len << 24);
    m.rec[r] = q;
}

struct Range {
    int first, end, stride;
};
__device__ __forceinline__ Range xcd_range(int nitems, int g, int wg)
{
    Range r;
    const int xcd = wg & 7, slot = wg >> 3;
    const int per = (nitems + 7) >> 3;
    r.first = xcd * per + slot;
    r.end = min((xcd + 1) * per, nitems);
    r.stride = g >> 3;
    return r;
}

__device__ __forceinline__ double block_sum(double v, double *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// VAR 0 = A, 1 = B
template <int VAR>
__global__ __launch_bounds__(256, 8) void k_cur(Mat m, const double *__restrict__ x, double *__restrict__ y, double sx,
                                                double sy, double cy, double *__restrict__ partials, int nblk, int persistent)
{
    __shared__ double red[4];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = VAR == 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    sdict[tid] = m.dict[tid];
    __syncthreads();
    double sq = 0.0;
    Range xr = persistent ? xcd_range(nblk, gridDim.x, blockIdx.x) : Range{(int)blockIdx.x, nblk, (int)gridDim.x};
    for (int b = xr.first; b < xr.end; b += xr.stride) {
        const int s = b * 4 + wave;
        if (s >= m.nslices) continue;
        unsigned o0, o1;
        int cb;
        if (VAR == 1) {
            o0 = m.soff[s];
            o1 = m.soff[s + 1];
            cb = m.cbase[s];
        } else {
            o0 = (unsigned)__builtin_amdgcn_readfirstlane((int)m.soff[s]);
            o1 = (unsigned)__builtin_amdgcn_readfirstlane((int)m.soff[s + 1]);
            cb = __builtin_amdgcn_readfirstlane(m.cbase[s]);
        }
        const int Ws = (int)((o1 - o0) >> 6);
        const int r = s * 64 + lane;
        const bool active = r < m.rows;
        const int rc = active ? r : m.rows - 1;
        const int len = active ? (int)m.rlen[rc] : 0;
        const double y0 = y[rc];
        double sum = 0.0;
        if (Ws == W) {
            int c[W], code[W];
            double xv[W];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                c[k] = cb + (int)m.sc16[(size_t)o0 + 64 * k + lane];
                code[k] = (int)m.sv8[(size_t)o0 + 64 * k + lane];
            }
#pragma unroll
            for (int k = 0; k < W; ++k) xv[k] = x[c[k]];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const double p = sdict[code[k]] * (xv[k] * sx);
                if (k < len) sum = sum + p;
            }
        }
        if (active) {
            const double yn = cy * (y0 * sy) + sum;
            y[r] = yn;
            sq += yn * yn;
        }
    }
    const double tot = block_sum(sq, red);
    if (tid == 0) partials[blockIdx.x] = tot;
}

struct Rec {
    uint4 q;
    double y0;
};

__device__ __forceinline__ double rec_sum(const uint4 q, int cb, const double *__restrict__ x, const double *sdict,
                                          double sx, bool gather, int r)
{
    int c[W], code[W];
    c[0] = cb + (int)(q.x & 0xffffu);
    c[1] = cb + (int)(q.x >> 16);
    c[2] = cb + (int)(q.y & 0xffffu);
    c[3] = cb + (int)(q.y >> 16);
    c[4] = cb + (int)(q.z & 0xffffu);
    code[0] = (int)((q.z >> 16) & 0xffu);
    code[1] = (int)(q.z >> 24);
    code[2] = (int)(q.w & 0xffu);
    code[3] = (int)((q.w >> 8) & 0xffu);
    code[4] = (int)((q.w >> 16) & 0xffu);
    const int len = (int)(q.w >> 24);
    double xv[W];
    if (gather) {
#pragma unroll
        for (int k = 0; k < W; ++k) xv[k] = x[c[k]];
    } else {
        const double t = x[r];
#pragma unroll
        for (int k = 0; k < W; ++k) xv[k] = t + (double)c[k];
    }
    double sum = 0.0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
        const double p = sdict[code[k]] * (xv[k] * sx);
        if (k < len) sum = sum + p;
    }
    return sum;
}

// the library's lazy-norm prologue: every workgroup reduces the previous kernel's 2048 partials
__device__ __forceinline__ double all_sum(const double *__restrict__ p, int np, double *red)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += p[i];
    const double r = block_sum(s, red);
    __syncthreads();
    if (threadIdx.x == 0) red[4] = r;
    __syncthreads();
    const double out = red[4];
    __syncthreads();
    return out;
}

// VAR 2 = C (one slice per wave trip), 3 = D (two slices in flight), 4 = S (no gather),
// 5 = C + stop-flag check first, 6 = C + stop flag + reduction of 2048 partials (lazy coefficients)
template <int VAR>
__global__ __launch_bounds__(256, 8) void k_rec(Mat m, const double *__restrict__ x, double *__restrict__ y, double sx,
                                                double sy, double cy, double *__restrict__ partials, int nblk, int persistent,
                                                const int *__restrict__ stop = nullptr,
                                                const double *__restrict__ pin = nullptr)
{
    __shared__ double red[5];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (VAR >= 5 && *stop != 0) return;
    sdict[tid] = m.dict[tid];
    if (VAR == 6) {
        const double nrm = all_sum(pin, 2048, red);
        sx = sx + 0.0 * nrm;   // the body depends on the reduced value, as in the library
    }
    __syncthreads();
    double sq = 0.0;
    Range xr = persistent ? xcd_range(nblk, gridDim.x, blockIdx.x) : Range{(int)blockIdx.x, nblk, (int)gridDim.x};
    const int last = m.nslices - 1;
    if (VAR == 3) {
        for (int b = xr.first; b < xr.end; b += 2 * xr.stride) {
            const int sa = b * 4 + wave;
            const int b2 = b + xr.stride;
            const int sbv = b2 * 4 + wave;
            const bool hasA = sa <= last, hasB = b2 < xr.end && sbv <= last;
            const int sA = min(sa, last), sB = min(sbv, last);
            const int rA = min(sA * 64 + lane, m.rows - 1), rB = min(sB * 64 + lane, m.rows - 1);
            const uint4 qa = m.rec[sA * 64 + lane];
            const uint4 qb = m.rec[sB * 64 + lane];
            const int cba = m.cbase[sA], cbb = m.cbase[sB];
            const double ya = y[rA], yb = y[rB];
            const double suma = rec_sum(qa, cba, x, sdict, sx, true, rA);
            const double sumb = rec_sum(qb, cbb, x, sdict, sx, true, rB);
            if (hasA && sa * 64 + lane < m.rows) {
                const double yn = cy * (ya * sy) + suma;
                y[rA] = yn;
                sq += yn * yn;
            }
            if (hasB && sbv * 64 + lane < m.rows) {
                const double yn = cy * (yb * sy) + sumb;
                y[rB] = yn;
                sq += yn * yn;
            }
        }
    } else {
        for (int b = xr.first; b < xr.end; b += xr.stride) {
            const int s = b * 4 + wave;
            if (s > last) continue;
            const int r = s * 64 + lane;
            const bool active = r < m.rows;
            const int rc = active ? r : m.rows - 1;
            const uint4 q = m.rec[s * 64 + lane];
            const int cb = m.cbase[s];
            const double y0 = y[rc];
            const double sum = rec_sum(q, cb, x, sdict, sx, VAR != 4, rc);
            if (active) {
                const double yn = cy * (y0 * sy) + sum;
                y[r] = yn;
                sq += yn * yn;
            }
        }
    }
    const double tot = block_sum(sq, red);
    if (tid == 0) partials[blockIdx.x] = tot;
}

// Workgroup shape: BS threads (BS/64 slices per trip), `np` = grid partials reduced by every
// workgroup when LAZY (fewer, fatter workgroups -> fewer partials to re-read).
template <int BS, bool LAZY>
__global__ __launch_bounds__(BS, 2048 / BS) void k_big(Mat m, const double *__restrict__ x, double *__restrict__ y,
                                                       double sx, double sy, double cy, double *__restrict__ partials,
                                                       int nblk, const int *__restrict__ stop,
                                                       const double *__restrict__ pin, int np)
{
    constexpr int NW = BS / 64;
    __shared__ double red[NW + 1];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (*stop != 0) return;
    if (tid < 256) sdict[tid] = m.dict[tid];
    if (LAZY) {
        double s = 0.0;
        for (int i = tid; i < np; i += BS) s += pin[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NW; ++w) t += red[w];
            red[NW] = t;
        }
        __syncthreads();
        sx = sx + 0.0 * red[NW];
    }
    __syncthreads();
    double sq = 0.0;
    const Range xr = xcd_range(nblk, gridDim.x, blockIdx.x);
    const int last = m.nslices - 1;
    for (int b = xr.first; b < xr.end; b += xr.stride) {
        const int s = b * NW + wave;
        if (s > last) continue;
        const int r = s * 64 + lane;
        const bool active = r < m.rows;
        const int rc = active ? r : m.rows - 1;
        const uint4 q = m.rec[s * 64 + lane];
        const int cb = m.cbase[s];
        const double y0 = y[rc];
        const double sum = rec_sum(q, cb, x, sdict, sx, true, rc);
        if (active) {
            const double yn = cy * (y0 * sy) + sum;
            y[r] = yn;
            sq += yn * yn;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += red[w];
        partials[blockIdx.x] = t;
    }
}

// Prologue anatomy.  PV: 0 none | 1 every thread loads its 8 partials, wave sum only (no barrier) |
// 2 the library's shape (loads + 3 barriers) | 3 only wave 0 loads (32 per lane), one barrier |
// 4 shape 2 on 64 partials only.  The partials read are the ones the PREVIOUS launch wrote
// (ping-pong), as in the solver: they come from beyond L2.
template <int PV>
__global__ __launch_bounds__(256, 8) void k_pro(Mat m, const double *__restrict__ x, double *__restrict__ y, double sx,
                                                double sy, double cy, double *__restrict__ pout, int nblk,
                                                const int *__restrict__ stop, const double *__restrict__ pin)
{
    __shared__ double red[5];
    __shared__ double sdict[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (*stop != 0) return;
    sdict[tid] = m.dict[tid];
    if (PV == 1) {
        double s = 0.0;
        for (int i = tid; i < 2048; i += 256) s += pin[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        sx = sx + 0.0 * s;
    } else if (PV == 2 || PV == 4) {
        const double nrm = all_sum(pin, PV == 4 ? 64 : 2048, red);
        sx = sx + 0.0 * nrm;
    } else if (PV == 3) {
        if (wave == 0) {
            double s = 0.0;
            for (int i = lane; i < 2048; i += 64) s += pin[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) red[4] = s;
        }
        __syncthreads();
        sx = sx + 0.0 * red[4];
    }
    __syncthreads();
    double sq = 0.0;
    const Range xr = xcd_range(nblk, gridDim.x, blockIdx.x);
    const int last = m.nslices - 1;
    for (int b = xr.first; b < xr.end; b += xr.stride) {
        const int s = b * 4 + wave;
        if (s > last) continue;
        const int r = s * 64 + lane;
        const bool active = r < m.rows;
        const int rc = active ? r : m.rows - 1;
        const uint4 q = m.rec[s * 64 + lane];
        const int cb = m.cbase[s];
        const double y0 = y[rc];
        const double sum = rec_sum(q, cb, x, sdict, sx, true, rc);
        if (active) {
            const double yn = cy * (y0 * sy) + sum;
            y[r] = yn;
            sq += yn * yn;
        }
    }
    const double tot = block_sum(sq, red);
    if (tid == 0) pout[blockIdx.x] = tot;
}

template <typename F>
static double time_us(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}

int main()
{
    for (int nx : {1000, 2000}) {
        const int ny = nx;
        Mat m{};
        m.rows = nx * ny;
        m.nslices = (m.rows + 63) / 64;
        const size_t ne = (size_t)m.nslices * 64 * W;
        CK(hipMalloc(&m.soff, 4 * (m.nslices + 1)));
        CK(hipMalloc(&m.cbase, 4 * m.nslices));
        CK(hipMalloc(&m.sc16, 2 * ne));
        CK(hipMalloc(&m.sv8, ne));
        CK(hipMalloc(&m.rlen, m.rows));
        CK(hipMalloc(&m.rec, 16 * (size_t)m.nslices * 64));
        CK(hipMalloc(&m.dict, 8 * 256));
        double hd[256] = {4.0, -1.0};
        CK(hipMemcpy(m.dict, hd, sizeof hd, hipMemcpyHostToDevice));
        double *x, *y, *ya, *partials;
        CK(hipMalloc(&x, 8 * (size_t)m.rows));
        CK(hipMalloc(&y, 8 * (size_t)m.rows));
        CK(hipMalloc(&ya, 8 * (size_t)m.rows));
        CK(hipMalloc(&partials, 8 * 65536));
        int *zero;
        double *pin;
        CK(hipMalloc(&zero, 64));
        CK(hipMemset(zero, 0, 64));
        CK(hipMalloc(&pin, 8 * 2048));
        CK(hipMemset(pin, 0, 8 * 2048));
        hipLaunchKernelGGL(k_build, dim3((m.nslices * 64 + 255) / 256), dim3(256), 0, 0, m, nx, ny);
        CK(hipDeviceSynchronize());
        double *hx = (double *)malloc(8 * (size_t)m.rows);
        for (int i = 0; i < m.rows; ++i) hx[i] = 1.0 + (i % 7) * 0.125;
        CK(hipMemcpy(x, hx, 8 * (size_t)m.rows, hipMemcpyHostToDevice));
        const int nblk = (m.nslices + 3) / 4;
        const double sx = 0.5, sy = 1.0, cy = 0.0;  // y = A (x/2): the same every launch
        // agreement A vs C
        CK(hipMemset(y, 0, 8 * (size_t)m.rows));
        CK(hipMemset(ya, 0, 8 * (size_t)m.rows));
        hipLaunchKernelGGL(k_cur<0>, dim3(2048), dim3(256), 0, 0, m, x, ya, sx, sy, cy, partials, nblk, 1);
        hipLaunchKernelGGL(k_rec<3>, dim3(2048), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, 1);
        CK(hipDeviceSynchronize());
        double *h1 = (double *)malloc(8 * (size_t)m.rows), *h2 = (double *)malloc(8 * (size_t)m.rows);
        CK(hipMemcpy(h1, ya, 8 * (size_t)m.rows, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2, y, 8 * (size_t)m.rows, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (int i = 0; i < m.rows; ++i) bad += h1[i] != h2[i];
        std::printf("\nPoisson %d x %d: %d rows, %.1f MB per product (40 B/row); A vs D mismatches: %zu\n", nx, ny, m.rows,
                    40e-6 * m.rows, bad);
        std::printf("%-52s %10s %10s\n", "variant", "us", "TB/s");
        const int reps = 200;
        for (int persistent : {1, 0}) {
            const int grid = persistent ? 2048 : nblk;
            const char *g = persistent ? "2048 wg" : "1 wg / 4 slices";
            auto rep = [&](const char *name, double us) {
                char buf[96];
                std::snprintf(buf, sizeof buf, "%s [%s]", name, g);
                std::printf("%-52s %10.2f %10.2f\n", buf, us, 40e-6 * m.rows / us);
            };
            rep("A  sell.h layout, vector descriptors", time_us([&] {
                    hipLaunchKernelGGL(k_cur<0>, dim3(grid), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, persistent);
                }, reps));
            rep("B  sell.h layout, scalar descriptors", time_us([&] {
                    hipLaunchKernelGGL(k_cur<1>, dim3(grid), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, persistent);
                }, reps));
            rep("C  16-byte records", time_us([&] {
                    hipLaunchKernelGGL(k_rec<2>, dim3(grid), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, persistent);
                }, reps));
            if (persistent)
                rep("D  16-byte records, two slices in flight", time_us([&] {
                        hipLaunchKernelGGL(k_rec<3>, dim3(grid), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, 1);
                    }, reps));
            if (persistent) {
                rep("C5 C + stop-flag load and branch first", time_us([&] {
                        hipLaunchKernelGGL(k_rec<5>, dim3(grid), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, 1,
                                           (const int *)zero, (const double *)pin);
                    }, reps));
                rep("C6 C5 + every workgroup reduces 2048 partials", time_us([&] {
                        hipLaunchKernelGGL(k_rec<6>, dim3(grid), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, 1,
                                           (const int *)zero, (const double *)pin);
                    }, reps));
            }
            rep("S  16-byte records, no gather (stream)", time_us([&] {
                    hipLaunchKernelGGL(k_rec<4>, dim3(grid), dim3(256), 0, 0, m, x, y, sx, sy, cy, partials, nblk, persistent);
                }, reps));
        }
        {
            double *pp[2];
            CK(hipMalloc(&pp[0], 8 * 2048));
            CK(hipMalloc(&pp[1], 8 * 2048));
            CK(hipMemset(pp[0], 0, 8 * 2048));
            CK(hipMemset(pp[1], 0, 8 * 2048));
            int flip = 0;
            auto rp = [&](const char *name, double us) { std::printf("%-60s %10.2f\n", name, us); };
#define PRO(PV, NAME)                                                                                                  \
    rp(NAME, time_us([&] {                                                                                             \
           hipLaunchKernelGGL(k_pro<PV>, dim3(2048), dim3(256), 0, 0, m, x, y, sx, sy, cy, pp[flip ^ 1], nblk,         \
                              (const int *)zero, (const double *)pp[flip]);                                           \
           flip ^= 1;                                                                                                  \
       }, reps));
            PRO(0, "prologue: none")
            PRO(1, "prologue: 8 loads per thread + wave sum, no barrier")
            PRO(2, "prologue: library shape (8 loads, 3 barriers), 2048 partials")
            PRO(4, "prologue: library shape, 64 partials")
            PRO(3, "prologue: wave 0 loads all 2048, one barrier")
#undef PRO
        }
        {
            auto rep = [&](const char *name, int bs, int grid, double us) {
                std::printf("%-36s %5d thr x %5d wg %10.2f %10.2f\n", name, bs, grid, us, 40e-6 * m.rows / us);
            };
#define BIG(BS, GRID)                                                                                                  \
    {                                                                                                                  \
        const int nb = (m.nslices + (BS) / 64 - 1) / ((BS) / 64);                                                      \
        rep("records, coefficients given", BS, GRID, time_us([&] {                                                     \
                hipLaunchKernelGGL((k_big<BS, false>), dim3(GRID), dim3(BS), 0, 0, m, x, y, sx, sy, cy, partials, nb,   \
                                   (const int *)zero, (const double *)pin, GRID);                                     \
            }, reps));                                                                                                 \
        rep("records, lazy (reduce grid partials)", BS, GRID, time_us([&] {                                            \
                hipLaunchKernelGGL((k_big<BS, true>), dim3(GRID), dim3(BS), 0, 0, m, x, y, sx, sy, cy, partials, nb,    \
                                   (const int *)zero, (const double *)pin, GRID);                                     \
            }, reps));                                                                                                 \
    }
            BIG(256, 2048) BIG(256, 1024) BIG(512, 1024) BIG(512, 512) BIG(1024, 512) BIG(1024, 256)
#undef BIG
        }
        for (void *p : {(void *)m.soff, (void *)m.cbase, (void *)m.sc16, (void *)m.sv8, (void *)m.rlen, (void *)m.rec,
                        (void *)m.dict, (void *)x, (void *)y, (void *)ya, (void *)partials})
            CK(hipFree(p));
        free(hx);
        free(h1);
        free(h2);
    }
    return 0;
}
