// grid_barrier.hip -- what does a phase boundary cost on one MI355X: a kernel launch inside a
// hipGraph, or a grid-wide barrier inside one persistent kernel?
//
// Config 2 (1M x 1M Poisson) runs two launches per LSQR iteration of ~40 + ~80 MB; ~3.7 us of
// each is fixed cost.  This measures the alternative: the same streaming phases
// (read 4 doubles, write 1 per element; n = 1M -> 40 MB per phase) separated by
//   (a) launch boundaries: R kernel nodes in a graph,
//   (b) a sense-counting grid barrier (device-scope release/acquire) in ONE launch,
// plus the bare barrier with no traffic.  Every spin is bounded: a barrier that does not
// complete in 2^22 polls sets an error flag and the kernel returns (no hang).
//
//   hipcc --offload-arch=gfx950 -O3 scripts/grid_barrier.hip -o gpurun_out/grid_barrier
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

struct Bar {
    unsigned count;
    unsigned pad[31];
    unsigned err;
};

// All threads of the workgroup call this.  `gen` counts barriers passed so far (same everywhere).
__device__ __forceinline__ bool grid_barrier(Bar *b, unsigned nwg, unsigned &gen)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned target = (gen + 1) * nwg;
        __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned polls = 0;
        while (__hip_atomic_load(&b->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++polls > (1u << 22)) {
                b->err = 1;
                ok = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    ++gen;
    __syncthreads();
    return ok;
}

__device__ __forceinline__ void phase(const double *__restrict__ a, const double *__restrict__ b,
                                      const double *__restrict__ c, const double *__restrict__ d,
                                      double *__restrict__ y, int n, int wg, int nwg)
{
    for (int i = wg * 256 + threadIdx.x; i < n; i += nwg * 256) y[i] = a[i] + b[i] * c[i] + d[i];
}

__global__ __launch_bounds__(256, 8) void k_phase(const double *a, const double *b, const double *c, const double *d,
                                                  double *y, int n)
{
    phase(a, b, c, d, y, n, blockIdx.x, gridDim.x);
}

template <bool TRAFFIC>
__global__ __launch_bounds__(256, 8) void k_persistent(const double *a, const double *b, const double *c,
                                                       const double *d, double *y0, double *y1, int n, int reps, Bar *bar)
{
    unsigned gen = 0;
    for (int r = 0; r < reps; ++r) {
        // ping-pong so that every phase reads what the previous one wrote somewhere else on the chip
        if (TRAFFIC) phase(a, b, c, (r & 1) ? y0 : y1, (r & 1) ? y1 : y0, n, blockIdx.x, gridDim.x);
        if (!grid_barrier(bar, gridDim.x, gen)) return;
    }
}

static float timed(hipStream_t st, void (*f)(hipStream_t, void *), void *ctx)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f(st, ctx);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    f(st, ctx);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

struct Ctx {
    double *a, *b, *c, *d, *y0, *y1;
    int n, reps, grid;
    Bar *bar;
    hipGraphExec_t ge;
    bool traffic;
};

static void run_graph(hipStream_t st, void *p) { CK(hipGraphLaunch(((Ctx *)p)->ge, st)); }
static void run_persist(hipStream_t st, void *p)
{
    Ctx *c = (Ctx *)p;
    CK(hipMemsetAsync(c->bar, 0, sizeof(Bar), st));
    if (c->traffic)
        hipLaunchKernelGGL(k_persistent<true>, dim3(c->grid), dim3(256), 0, st, c->a, c->b, c->c, c->d, c->y0, c->y1, c->n,
                           c->reps, c->bar);
    else
        hipLaunchKernelGGL(k_persistent<false>, dim3(c->grid), dim3(256), 0, st, c->a, c->b, c->c, c->d, c->y0, c->y1,
                           c->n, c->reps, c->bar);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_persistent<true>, 256, 0));
    std::printf("CUs %d, resident workgroups per CU (256 threads) %d\n", cus, per_cu);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Ctx c{};
    c.n = 1000000;
    c.reps = 400;
    for (double **p : {&c.a, &c.b, &c.c, &c.d, &c.y0, &c.y1}) {
        CK(hipMalloc(p, 8 * (size_t)c.n));
        CK(hipMemset(*p, 0, 8 * (size_t)c.n));
    }
    CK(hipMalloc(&c.bar, sizeof(Bar)));

    std::printf("%-44s %8s %12s\n", "variant", "grid", "us / phase");
    for (int mult : {1, 2, 4, 8}) {
        if (mult > per_cu) break;
        c.grid = cus * mult;
        // (a) launches in a graph
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int r = 0; r < c.reps; ++r)
            hipLaunchKernelGGL(k_phase, dim3(c.grid), dim3(256), 0, st, c.a, c.b, c.c, (r & 1) ? c.y0 : c.y1,
                               (r & 1) ? c.y1 : c.y0, c.n);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&c.ge, g, nullptr, nullptr, 0));
        const float tg = timed(st, run_graph, &c);
        std::printf("%-44s %8d %12.2f\n", "40 MB phase, launch per phase (graph)", c.grid, 1e3 * tg / c.reps);
        CK(hipGraphExecDestroy(c.ge));
        CK(hipGraphDestroy(g));
        // (b) persistent + barrier
        c.traffic = true;
        const float tp = timed(st, run_persist, &c);
        std::printf("%-44s %8d %12.2f\n", "40 MB phase, persistent + grid barrier", c.grid, 1e3 * tp / c.reps);
        c.traffic = false;
        const float tb = timed(st, run_persist, &c);
        std::printf("%-44s %8d %12.2f\n", "bare grid barrier", c.grid, 1e3 * tb / c.reps);
        unsigned err = 0;
        CK(hipMemcpy(&err, &c.bar->err, 4, hipMemcpyDeviceToHost));
        if (err) std::printf("  !! a barrier timed out at grid %d\n", c.grid);
    }
    return 0;
}
