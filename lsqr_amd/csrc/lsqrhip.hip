// lsqrhip.hip -- liblsqrhip.so: host orchestration + C-ABI (include/lsqrhip.h).
//
// Drop-in for the reference's lsqr_solver_ez path:
//   lsqrhip_create  <- initialize_ez   (src/lsqr.f90:91-127)
//   lsqrhip_solve   <- solve_ez + LSQR (src/lsqr.f90:207-259, 432-882)
//   lsqrhip_aprod   <- aprod_ez        (src/lsqr.f90:134-200)
// The whole iteration runs device-resident: per iteration three HBM-bound vector
// kernels (mode-1 SpMV, mode-2 SpMV, x/w update) and three one-workgroup scalar
// kernels, captured in batches into a hipGraph; the host only polls the stop flag
// between batches.  There is no CPU fallback anywhere in this file.
#include "../../include/lsqrhip.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "csb.h"
#include "csr_build.h"
#include "scalar.h"
#include "sell.h"
#include "pat.h"
#include "spmv.h"
#include "xl.h"
#include "state.h"
#include "vec.h"

using namespace lsqrhip;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;
static std::atomic<int> g_device{0};  // device new handles are created on (process-wide, like hipSetDevice's default)

static int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            const int _c = (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice)            \
                               ? LSQRHIP_ERR_NO_DEVICE                                        \
                               : (_e == hipErrorOutOfMemory ? LSQRHIP_ERR_ALLOC : LSQRHIP_ERR_HIP); \
            return fail(_c, std::string(#expr) + ": " + hipGetErrorString(_e));               \
        }                                                                                     \
    } while (0)

#define RET(expr)                    \
    do {                             \
        int _rc = (expr);            \
        if (_rc != LSQRHIP_OK) return _rc; \
    } while (0)

extern "C" const char *lsqrhip_error_string(int code)
{
    switch (code) {
    case LSQRHIP_OK: return "ok";
    case LSQRHIP_ERR_SIZES: return "invalid a,icol,irow sizes in initialize_ez";
    case LSQRHIP_ERR_IROW: return "invalid irow or m in initialize_ez";
    case LSQRHIP_ERR_ICOL: return "invalid icol or n in initialize_ez";
    case LSQRHIP_ERR_NOT_INIT: return "lsqr_solver_ez class not properly initialized";
    case LSQRHIP_ERR_MODE: return "invalid mode in aprod_ez";
    case LSQRHIP_ERR_NO_DEVICE: return "no usable gfx950 (MI355X) device; the HIP path has no CPU fallback";
    case LSQRHIP_ERR_HIP: return "HIP runtime error";
    case LSQRHIP_ERR_ALLOC: return "device memory allocation failed";
    case LSQRHIP_ERR_ARG: return "invalid argument";
    case LSQRHIP_ERR_TOO_LARGE: return "problem too large for this build (nnz must be < 2^32)";
    default: return "unknown lsqrhip status";
    }
}

extern "C" const char *lsqrhip_last_error(void) { return g_last_error.c_str(); }

extern "C" int lsqrhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

extern "C" int lsqrhip_set_device(int device)
{
    if (device < 0) return fail(LSQRHIP_ERR_ARG, "negative device");
    g_device = device;
    return LSQRHIP_OK;
}

// The device of the handles THIS thread creates next, when >= 0 (lsqrhip_create_sharded places its row blocks
// with it); otherwise the process-wide selection of lsqrhip_set_device.
static thread_local int t_shard_world = 0;   // lsqrhip_create_sharded: the world its row blocks are being created for
static thread_local int t_device_override = -1;
static int target_device() { return t_device_override >= 0 ? t_device_override : g_device.load(); }

static int use_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LSQRHIP_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    const int dev = target_device();
    if (dev >= n) return fail(LSQRHIP_ERR_NO_DEVICE, "selected device index out of range");
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, dev));
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return fail(LSQRHIP_ERR_NO_DEVICE, std::string("device is ") + p.gcnArchName + ", kernels are built for gfx950 only");
    HIPCHK(hipSetDevice(dev));
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------
struct Csr {
    void *rowptr = nullptr;  // int32 or int64 [rows+1]
    int *col = nullptr;              // 32-bit column indices (freed when col16 is in use)
    unsigned short *col16 = nullptr;  // 16-bit block-relative column indices (spmv.h C16) ...
    int *cbase = nullptr;            // ... and each row block's smallest column
    double *val = nullptr;           // 8-byte values (freed when val8 is in use)
    unsigned char *val8 = nullptr;   // one-byte codes into dict (valdict.h) ...
    const double *dict = nullptr;    // ... the handle's dictionary (not owned)
    // sliced-ELL layout (sell.h), used instead of the arrays above when `sell` is set
    int sell = 0;                    // 1 = column-major slices, 2 = packed 16-byte records (sell.h), 3 = row patterns, 4 = structure patterns (pat.h)
    unsigned char *pid = nullptr;    // sell = 3: [rows] pattern of each row ...
    unsigned *pdesc = nullptr;       // ... [PAT_MAX] first entry | length << 16 of each pattern
    int *pdelta = nullptr;           // ... [PAT_MAX_E] column - row of each entry
    double *pval = nullptr;          // ... [PAT_MAX_E] value of each entry
    int npat = 0, npat_e = 0;        // patterns, entries in use
    bool pat_pair = false;           // sell = 3, one-byte table: lane L owns rows 2L, 2L + 1 of a 128-row group (pat.h "paired rows"); nblk counts blocks of 4 groups
    bool pat_wide = false;           // sell = 3, 257 ... 4096 patterns (pat.h "wide"): pid holds u16, the table is pent and stays in global memory
    int pat_u = 2;                   // ... slices a wave takes through a trip together (LSQRHIP_PAT2_U)
    void *pent = nullptr;            // ... [npat * pat_stride] PatEnt { value, column - row, length }: entry k of pattern p at p * pat_stride + k
    int pat_stride = 1;              // ... the longest pattern
    bool nt = false;                 // short-row layouts: the matrix stream is loaded non-temporally (common.h ld_stream)
    unsigned *soff = nullptr;        // [nslices+1] first element (sell = 2: first record) of each 64-row slice
    uint4 *srec = nullptr;           // sell = 2: records (5 columns, 5 value codes, row length)
    void *scol = nullptr;            // column-major columns (u16 relative to cbaseS, or i32)
    void *sval = nullptr;            // column-major values (u8 dictionary codes, or f64)
    int *cbaseS = nullptr;           // [nslices] smallest column of each slice
    unsigned char *rlen = nullptr;   // [rows] row lengths
    int nslices = 0;
    bool sell_c16 = false, sell_v8 = false;
    int *rb = nullptr;  // row-block boundaries [nblk+1]
    RowBlock *blk = nullptr;  // one descriptor per row block (spmv.h)
    int64_t nblk = 0;
    int rows = 0, cols = 0;
    int grid = 0;      // workgroups of the SpMV launch
    int out_grid = 0;  // partials one product leaves behind (== grid, or the combine kernel's grid)
    int P = 1;         // column panels (1 = plain CSR)
    int xlds = 0;      // LDS-resident x slices: 1 = spmv.h XL (256-thread), 2 = xl.h (1024-thread workgroups)
    int xgrid = 0;     // grid of the xl.h kernel
    int *gpid = nullptr;  // xl.h: panel of every trip of XLW_WAVES windows
    unsigned short *rel16 = nullptr;  // xl.h: 16-bit row starts relative to their window (rowptr is then released)
    unsigned char *skew = nullptr;  // spmv.h: windows of a panelled matrix that hold a segment > SPMV_LONGCUT
    int pw = 0;        // panel width in columns
    int64_t rows_v = 0;  // virtual rows = P * rows (what rowptr / rb / blk index)
    int64_t bytes = 0;
    int64_t nstored = 0; // value slots of the layout in use (nnz; padded elements of sliced ELL; csb chunks * 256)
    // column-swept row blocks (csb.h), used instead of everything above when `csb` is set
    int csb = 0;
    double *cval = nullptr;       // [nchunks * 256] values, each block sorted by column, padded to whole chunks
    unsigned *cidx = nullptr;     // [nchunks * 256] local row << 17 | column - cbase[chunk]; narrow form (csb.h): the u16 rows
    unsigned *cdel = nullptr;     // narrow form: [nchunks * 64] four u8 column deltas per lane
    int *ccb = nullptr;           // [nchunks] first column of each chunk; narrow form: [nchunks * 4] of each segment
    bool cnarrow = false;         // 11 bytes per nonzero (csb.h "NARROW form")
    int crounds = 1;              // one launch per round of 256 units (1) or one launch over all of them (0): LSQRHIP_CSB_ROUNDS
    int cbarrier_a = 0;           // lock step: the second barrier, in front of the gathers (LSQRHIP_CSB_BARRIER_A; by shape)
    int cstagger = 0;             // lock step: late start of every other workgroup of an XCD, x 2048 cycles (LSQRHIP_CSB_STAGGER)
    int clockstep = 2;            // chunks per wave and lock-step step of the sweep (csb.h "lock step"); 0: free-running waves
    // overlap plan of the sharded engine (csb.h "Column stripes / phases"); all off: NS = 1, border = null
    long long *gptr = nullptr;    // [nrb * NS + 1] first chunk of every (block, stripe) group
    int *border = nullptr;        // [nrb] launch order of the row blocks (phase-major), or null
    int NS = 1, G = 1, J = 1, Pst = 1;
    int phases = 1;               // launches of a product fall into this many phases (shard_engine.h waits between them)
    std::vector<int> phase_pos;   // border form: [phases + 1] positions of the launch order where the phases begin
    long long *cptr = nullptr;    // [nrb + 1] first chunk of each row block
    int *crs = nullptr;           // [nrb + 1] first row of each row block
    int64_t nchunks = 0;
    int nrb = 0, R = 0;
    short *rexp = nullptr;        // [rows] e1_i, 2^e1_i > the row's 1-norm: the stored values are a_ij 2^-e1_i (csb.h)
    long long *zcoarse = nullptr; // [rows] integer sums of the products with big columns (csb.h); zero between products
    int S = 1;                    // column splits per row block (csb.h): S workgroups share a block
    long long *zsplit = nullptr;  // S > 1: [S][rows] exact integer sums of the splits
    int *cbad = nullptr;          // S > 1: [nrb][CSB_QMAX] "a split left a product out / used the coarse sums" flags (csb.h)
    int Q = 1;                    // S > 1: workgroups of k_csb_combine per row block
    void *chand = nullptr;        // S > 1 with the combine launch: the sweeps' coefficients and grids for k_csb_combine (csb.h CsbHand)
    unsigned long long *cprobe = nullptr;   // LSQRHIP_CSB_PROBE=1 (measurement only): [CSB_PROBE_LAUNCHES][CSB_PROBE_WGS][8] phase clocks (csb.h)
    int cfuse = 0;                // S > 1: no k_csb_combine launch -- the split of a block that arrives LAST runs the block's epilogue
                                  // itself (csb.h "the last split closes the block"; LSQRHIP_CSB_FUSE=0: the combine launch of rounds 2-5)
};

// One rank's view of a row-sharded solve (shard_api.h): its place in the world, the caller-owned
// exchange buffers of the current solve, and two device words of its own.
struct ShardCtx {
    int P = 1, rank = 0;
    int64_t chunk = 0;            // ceil(n / P): columns per slice
    int64_t my0 = 0, mylen = 0;   // this rank's column slice [my0, my0 + mylen)
    double *T = nullptr, *R = nullptr, *V = nullptr, *sums = nullptr;  // caller-owned: P*chunk, P*chunk, P*chunk, 4
    double *wsq = nullptr;        // [1] this rank's sum of w_q^2 (owned)
    int *live = nullptr;          // [1] "this iteration runs" (owned)
    int wantse = 0;
    const double *gath = nullptr; // the C++ engine: where the all-gather of the norms lands (P messages of `msg` doubles) -- the
    int msg = 4;                  // scalar steps then sum the ranks themselves (shard_api.h k_shard_s1g / s2g / s3w)
    bool own_in_T = false;        // the rank's own slice of T is read in place by k_rs_combine (never copied to R)
    int upar = 0;                 // the C++ engine: which set of u's piece maxima (H::MXU) the last mode 1 raised
    bool vmax_msg = false;        // the caller's `sums` is a SHARD_MSG-double message that carries this rank's piece maxima of |v_q|
                                  // (shard_engine.h, which hands the gathered maxima to mode 1 in xmax_part)
    int want_log = 0;             // option "shard_log": this rank keeps the iteration log of the next sharded solves
                                  // (rank 0's business: the scalars are replicated and x(1) lies on its slice)
    bool active = false;
    bool fuse_s2 = true;          // LSQRHIP_SHARD_FUSE_S2, read once per solve in lsqrhip_shard_begin: scalar step 2 inside the update's launch
    bool engine_next = false;     // set by shard_engine.h right before ITS lsqrhip_shard_begin: the engine-only fields above
                                  // (gath, msg, own_in_T, vmax_msg) are meant.  Any other caller of lsqrhip_shard_begin --
                                  // the Python stage driver, also as the fall-back after an engine solve that FAILED half
                                  // way and never reached lsqrhip_shard_end -- finds them cleared there (round-4 advisor).
};

struct ShardGroup;  // shard_engine.h: the ranks of a sharded solve driven from this process (RCCL)
struct lsqrhip_handle_s;
static int64_t shard_effective(const lsqrhip_handle_s *h, int what);   // shard_engine.h

struct lsqrhip_handle_s {
    std::atomic<int> refs{1};  // lsqrhip_retain / lsqrhip_destroy
    int device = 0;
    int m = 0, n = 0;
    int64_t nnz = 0;
    bool off64 = false;
    int64_t build_peak = 0;      // peak device bytes of the create beyond the caller's triplets; kept: bytes it holds now
    int64_t build_kept = 0;
    bool f32_device_ok = false;  // (internal: lsqrhip_aprod_f32 calls lsqrhip_aprod_device on float vectors)
    bool io32 = false;  // created by lsqrhip_create_f32 (whether stored as float or, mixed mode, as double)
    bool f32 = false;   // REAL32 handle: values and the vectors U, V, W, X, SE are float arrays (the pointers below are
                        // then float* in disguise); arithmetic in registers stays binary64 (lsqrhip_create_f32)
    Csr A, AT;
    double *dict = nullptr;  // <= 256 distinct values of the matrix, ascending bit patterns (valdict.h)
    int ndict = 0;           // 0 = no dictionary
    double *U = nullptr, *V = nullptr, *W = nullptr, *X = nullptr, *SE = nullptr;
    double *Z = nullptr;         // per-panel row sums of a panelled product (max over A, A')
    double *partials = nullptr;  // 3 * SPMV_MAX_GRID (three planes for Blue's norm of b, vec.h k_sumsq3)
    double *xmax_part = nullptr; // piece maxima of |x| for csb.h products (k_csb_xmax; the sharded engine's gathered maxima of v)
    double *MXU = nullptr, *MXV = nullptr;  // piece maxima of U / V left by the column-swept product that wrote them (csb.h ymax)
    NScale nsc{1.0, 1.0};        // fused norms: sum of (y * nsc.s)^2, sqrt(sum) * nsc.inv (scalar.h "range-safe norms")
    int norm_exp = 0;            // nsc.s = 2^-norm_exp (option "norm_exp": ranks of a sharded solve agree on one)
    int amax_exp = 0;            // 2^amax_exp > max|a_ij| of THIS matrix (csb.h's bound on the products)
    int vgrid_m = 1, vgrid_n = 1;
    LsqrState *d_state = nullptr;
    const void **h_bslot = nullptr;  // pinned: where this solve's b lies (k_start reads it through here: vec.h)
    LsqrState *h_state = nullptr;  // pinned; [1], [2] = per-batch snapshots of the look-ahead poll (solve_loop.h)
    SpmvCoef *d_unit = nullptr;    // (1,1,1): plain y += A x
    int *d_zero = nullptr;         // a stop flag that is never set
    double *d_scalar = nullptr;    // 4 doubles scratch
    double *d_log = nullptr;
    int log_cap = 0;
    std::vector<double> h_log;
    int log_count = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // options
    int use_graph = 1, graph_iters = 64, time_kernels = 0;
    int poll_ahead = 1;  // enqueue the next graph batch before waiting on the current one
    bool loop_bracketed = false;  // the last solve recorded ev_loop0 / ev_loop1
    int loop_events = 0;  // HIP events around the iteration loop: timing.loop_ms (0 without them)
    hipEvent_t ev_batch[2] = {nullptr, nullptr};
    hipGraphExec_t gexec = nullptr;
    hipGraphExec_t gexec_first = nullptr;  // the start of a solve (state upload, memsets, the five initial kernels) + the first batch
    int gexec_first_wantse = -1;
    int gexec_iters = 0;
    bool graph_dirty = true;
    int graph_epoch = 0;  // bumped by whatever invalidates captured kernel nodes (log buffer moved, norm_exp changed): the
                          // sharded engine's graphs (shard_engine.h) are rebuilt when it differs from the one they were captured at
    std::vector<hipEvent_t> ev;
    hipEvent_t ev_loop0 = nullptr, ev_loop1 = nullptr;
    lsqrhip_timing_t timing{};
    // rider schedule (solve_loop.h)
    int pipeline = 2;  // 0 sequential | 1 riders | 2 riders + x/w update fused into mode 1 (solve_loop.h)
    int gexec_pipeline = -1;
    double *P1[2] = {nullptr, nullptr};  // mode-1 partials by iteration parity
    double *P2[2] = {nullptr, nullptr};  // mode-2 partials by iteration parity
    double *P3 = nullptr;                // x/w update partials
    NormSlot *slots = nullptr;           // [0..1] alpha side (from mode 1), [2..3] beta side (from mode 2)
    // user device operator (op_api.h); A / AT are empty for such a handle
    lsqrhip_aprod_fn op = nullptr;
    void *op_user = nullptr;
    void (*op_free)(void *) = nullptr;  // releases op_user with the handle (built-in operators)
    bool op_is_lstp = false;
    double *opX = nullptr, *opY = nullptr;  // scaled copies handed to the operator (n, m)
    int op_batch = 8;                       // iterations enqueued ahead of each stop poll
    // row-sharded solve
    ShardCtx shard;                // this handle as one rank (shard_api.h)
    ShardGroup *group = nullptr;   // this handle as a whole sharded system, one process (shard_engine.h); owned
    ShardGroup *mp = nullptr;      // this handle as ONE rank of a multi-process world (shard_engine.h); owned
};
typedef lsqrhip_handle_s H;

// Device scratch owned by a scope: freed on every exit path (an allocation that fails half way
// through a build must not strand the buffers before it).
struct DevScratch {
    void *p = nullptr;
    DevScratch() = default;
    DevScratch(const DevScratch &) = delete;
    DevScratch &operator=(const DevScratch &) = delete;
    ~DevScratch()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes > 0 ? bytes : 1); }
    void free_now()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    template <typename T>
    T *as() const
    {
        return static_cast<T *>(p);
    }
    template <typename T>
    T *release()  // ownership moves to the caller
    {
        T *q = static_cast<T *>(p);
        p = nullptr;
        return q;
    }
};

// Peak device memory of a create beyond what the caller holds (the COO triplets): free memory is sampled where the
// build's footprint peaks (after its scratch and the layout arrays are allocated) -- get_option "build_peak_bytes".
static thread_local size_t t_min_free = ~(size_t)0;
static void probe_mem()
{
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < t_min_free) t_min_free = fr;
}

static int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

static void dbg_stage(hipStream_t s, const char *what)
{
    static const int on = env_int("LSQRHIP_TRACE", 0);
    if (!on) return;
    hipError_t e = hipStreamSynchronize(s);
    std::fprintf(stderr, "[lsqrhip trace] %s: %s\n", what, hipGetErrorString(e));
    std::fflush(stderr);
}

static int vec_grid(int64_t n)
{
    int64_t g = (n / 2 + VEC_BLOCK - 1) / VEC_BLOCK;
    if (g < 1) g = 1;
    if (g > VEC_MAX_GRID) g = VEC_MAX_GRID;
    return (int)g;
}

static void free_csr(Csr &c)
{
    if (c.rowptr) (void)hipFree(c.rowptr);
    if (c.col) (void)hipFree(c.col);
    if (c.col16) (void)hipFree(c.col16);
    if (c.cbase) (void)hipFree(c.cbase);
    if (c.val) (void)hipFree(c.val);
    if (c.val8) (void)hipFree(c.val8);
    if (c.soff) (void)hipFree(c.soff);
    if (c.skew) (void)hipFree(c.skew);
    if (c.rel16) (void)hipFree(c.rel16);
    if (c.srec) (void)hipFree(c.srec);
    if (c.scol) (void)hipFree(c.scol);
    if (c.sval) (void)hipFree(c.sval);
    if (c.cbaseS) (void)hipFree(c.cbaseS);
    if (c.rlen) (void)hipFree(c.rlen);
    if (c.pid) (void)hipFree(c.pid);
    if (c.pdesc) (void)hipFree(c.pdesc);
    if (c.pdelta) (void)hipFree(c.pdelta);
    if (c.pval) (void)hipFree(c.pval);
    if (c.pent) (void)hipFree(c.pent);
    if (c.rb) (void)hipFree(c.rb);
    if (c.gpid) (void)hipFree(c.gpid);
    if (c.blk) (void)hipFree(c.blk);
    if (c.cval) (void)hipFree(c.cval);
    if (c.cidx) (void)hipFree(c.cidx);
    if (c.cdel) (void)hipFree(c.cdel);
    if (c.gptr) (void)hipFree(c.gptr);
    if (c.border) (void)hipFree(c.border);
    if (c.ccb) (void)hipFree(c.ccb);
    if (c.cptr) (void)hipFree(c.cptr);
    if (c.crs) (void)hipFree(c.crs);
    if (c.zsplit) (void)hipFree(c.zsplit);
    if (c.rexp) (void)hipFree(c.rexp);
    if (c.zcoarse) (void)hipFree(c.zcoarse);
    if (c.cbad) (void)hipFree(c.cbad);
    if (c.cprobe) (void)hipFree(c.cprobe);
    if (c.chand) (void)hipFree(c.chand);
    c = Csr();
}

static void release_groups(H *h);  // shard_engine.h
static int solve_group_host(H *h, const double *b, double damp, double atol, double btol, double conlim, int itnlim,
                            int wantse, int want_log, double *x, double *se, int *istop, int *itn, double *anorm,
                            double *acond, double *rnorm, double *arnorm, double *xnorm);
static int aprod_group_host(H *h, int mode, double *x, double *y);
static int aprod_group_host_f32(H *h, int mode, float *x, float *y);

static H *lsqrhip_group_rank0(H *h);

static void destroy_graph(H *h)
{
    if (h->gexec) {
        (void)hipGraphExecDestroy(h->gexec);
        h->gexec = nullptr;
    }
    if (h->gexec_first) {
        (void)hipGraphExecDestroy(h->gexec_first);
        h->gexec_first = nullptr;
    }
    h->graph_dirty = true;
}

extern "C" int lsqrhip_retain(lsqrhip_handle_t h)
{
    if (!h) return fail(LSQRHIP_ERR_ARG, "null handle");
    ++h->refs;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_destroy(lsqrhip_handle_t h)
{
    if (!h) return LSQRHIP_OK;
    if (--h->refs > 0) return LSQRHIP_OK;
    release_groups(h);
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    destroy_graph(h);
    free_csr(h->A);
    free_csr(h->AT);
    for (double *p : {h->U, h->V, h->W, h->X, h->SE, h->Z, h->partials, h->xmax_part, h->MXU, h->MXV, h->d_scalar, h->d_log,
                      h->dict, h->opX, h->opY})
        if (p) (void)hipFree(p);
    if (h->op_free) h->op_free(h->op_user);
    if (h->shard.wsq) (void)hipFree(h->shard.wsq);
    if (h->shard.live) (void)hipFree(h->shard.live);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->h_bslot) (void)hipHostFree((void *)h->h_bslot);
    if (h->d_unit) (void)hipFree(h->d_unit);
    if (h->d_zero) (void)hipFree(h->d_zero);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (double *p : {h->P1[0], h->P1[1], h->P2[0], h->P2[1], h->P3})
        if (p) (void)hipFree(p);
    if (h->slots) (void)hipFree(h->slots);
    for (hipEvent_t e : h->ev_batch)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_loop0) (void)hipEventDestroy(h->ev_loop0);
    if (h->ev_loop1) (void)hipEventDestroy(h->ev_loop1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// K0: build one CSR from device COO (keys = row index of the CSR being built)
// ---------------------------------------------------------------------------
static int bits_for(int limit)
{
    int b = 1;
    while (b < 31 && (1ll << b) < (long long)limit) ++b;
    return b;
}

// Sliced-ELL layout for short, even rows (sell.h).  On success out.sell = 1 and the CSR arrays
// col / val are released; otherwise `out` is left as it was.  `stats` is scratch of >= 4 words.
static void launch_scan_small(hipStream_t s, unsigned *a, int64_t L)
{
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, a, L);
}

// Row patterns (pat.h): a matrix with <= 256 distinct rows keeps one byte per row.  On success out.sell = 3 and the
// CSR arrays col / val are released; otherwise `out` is left as it was.
static int try_pat(hipStream_t s, Csr &out, int64_t nnz, bool vals, unsigned long long *stats, bool *too_many = nullptr)
{
    dbg_stage(s, vals ? "try_pat(vals) enter" : "try_pat(structure) enter");
    const int rows = out.rows;
    // vals = false: structure patterns (pat.h "sell = 4") -- the column structure of the rows repeats, their values
    // do not; LSQRHIP_SPAT=0 never, =1 whenever the limits hold
    const int mode = vals ? env_int("LSQRHIP_PAT", -1) : env_int("LSQRHIP_SPAT", -1);
    if (out.P > 1 || nnz <= 0 || rows <= 0 || mode == 0) return LSQRHIP_OK;
    if (nnz > (int64_t)PAT_MAX_LEN * rows) return LSQRHIP_OK;
    DevScratch s_keys, s_reps, s_slot, s_ctl, s_desc, s_delta, s_val, s_pid;
    HIPCHK(s_keys.alloc(sizeof(unsigned long long) * PAT_TAB));
    HIPCHK(s_reps.alloc(sizeof(int) * PAT_TAB));
    HIPCHK(s_slot.alloc(sizeof(int) * PAT_TAB));
    HIPCHK(s_ctl.alloc(sizeof(int) * 4));
    HIPCHK(hipMemsetAsync(s_keys.p, 0, sizeof(unsigned long long) * PAT_TAB, s));
    HIPCHK(hipMemsetAsync(s_reps.p, 0x7f, sizeof(int) * PAT_TAB, s));
    HIPCHK(hipMemsetAsync(s_ctl.p, 0, sizeof(int) * 4, s));
    const int g = (int)std::min<int64_t>(((int64_t)rows + 255) / 256, 2048);
    int *ctl = s_ctl.as<int>();
    hipLaunchKernelGGL(k_pat_discover, dim3(g), dim3(256), 0, s, (const int *)out.rowptr, (const int *)out.col,
                       (const double *)out.val, rows, vals ? 1 : 0, s_keys.as<unsigned long long>(), s_reps.as<int>(),
                       ctl);
    HIPCHK(hipGetLastError());
    dbg_stage(s, "k_pat_discover");
    int got[4];
    HIPCHK(hipMemcpyAsync(got, ctl, sizeof(got), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (got[1] != 0 || got[0] > PAT_MAX) {
        if (too_many) *too_many = got[0] > PAT_MAX;   // (the 257th distinct row stopped the pass: the wide table may hold them)
        return LSQRHIP_OK;
    }
    if (mode != 1 && (int64_t)got[0] * 16 > rows) return LSQRHIP_OK;  // too few rows per pattern to be a structure
    HIPCHK(s_desc.alloc(sizeof(unsigned) * PAT_MAX));
    HIPCHK(s_delta.alloc(sizeof(int) * PAT_MAX_E));
    HIPCHK(s_val.alloc(sizeof(double) * PAT_MAX_E));
    HIPCHK(s_pid.alloc((size_t)rows + 2));   // (two bytes of padding: the paired-rows kernel reads two pattern numbers at once)
    HIPCHK(hipMemsetAsync(s_pid.as<unsigned char>() + rows, 0, 2, s));
    HIPCHK(hipMemsetAsync(s_desc.p, 0, sizeof(unsigned) * PAT_MAX, s));
    HIPCHK(hipMemsetAsync(s_slot.p, 0xff, sizeof(int) * PAT_TAB, s));
    HIPCHK(hipMemsetAsync(s_delta.p, 0, sizeof(int) * PAT_MAX_E, s));
    HIPCHK(hipMemsetAsync(s_val.p, 0, sizeof(double) * PAT_MAX_E, s));
    hipLaunchKernelGGL(k_pat_table, dim3(1), dim3(PAT_TAB), 0, s, (const int *)out.rowptr, (const int *)out.col,
                       (const double *)out.val, vals ? 1 : 0, (const unsigned long long *)s_keys.p, (const int *)s_reps.p,
                       s_slot.as<int>(), s_desc.as<unsigned>(), s_delta.as<int>(), s_val.as<double>(), ctl);
    hipLaunchKernelGGL(k_pat_assign, dim3(g), dim3(256), 0, s, (const int *)out.rowptr, (const int *)out.col,
                       (const double *)out.val, rows, vals ? 1 : 0, (const unsigned long long *)s_keys.p,
                       (const int *)s_slot.p,
                       (const unsigned *)s_desc.p, (const int *)s_delta.p, (const double *)s_val.p,
                       s_pid.as<unsigned char>(), ctl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(got, ctl, sizeof(got), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (got[1] != 0) return LSQRHIP_OK;   // too many entries, or two different rows under one key
    if (!vals) {   // structure patterns: the values stay, column-major per 64-row slice (sell.h's slices)
        const int nslices = (rows + 63) / 64;
        const unsigned gr = (unsigned)(((int64_t)nslices * 64 + 255) / 256);
        DevScratch s_off, s_sv;
        HIPCHK(s_off.alloc(sizeof(unsigned) * ((size_t)nslices + 1)));
        unsigned *soff = s_off.as<unsigned>();
        HIPCHK(hipMemsetAsync(stats, 0, 4 * sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync(soff + nslices, 0, sizeof(unsigned), s));
        hipLaunchKernelGGL(k_sell_width, dim3(gr), dim3(256), 0, s, (const int *)out.rowptr, rows, nslices, soff, stats);
        HIPCHK(hipGetLastError());
        unsigned long long st[4];
        HIPCHK(hipMemcpyAsync(st, stats, sizeof(st), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        const unsigned long long padded = st[0];
        if (padded > (unsigned long long)(nnz + nnz / 8 + 4096) || padded >= (1ull << 31)) return LSQRHIP_OK;
        launch_scan_small(s, soff, (int64_t)nslices + 1);
        const size_t np = (size_t)std::max<unsigned long long>(padded, 1);
        HIPCHK(s_sv.alloc(np * sizeof(double)));
        hipLaunchKernelGGL(k_spat_fill, dim3(gr), dim3(256), 0, s, (const int *)out.rowptr, (const double *)out.val,
                           (const unsigned *)soff, rows, nslices, s_sv.as<double>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        (void)hipFree(out.col);
        (void)hipFree(out.val);
        out.col = nullptr;
        out.val = nullptr;
        out.sell = 4;
        out.pid = s_pid.release<unsigned char>();
        out.pdesc = s_desc.release<unsigned>();
        out.pdelta = s_delta.release<int>();
        out.npat = got[2];
        out.npat_e = got[3];
        out.soff = s_off.release<unsigned>();
        out.sval = s_sv.release<void>();
        out.nslices = nslices;
        out.nblk = (nslices + SELL_SLICES - 1) / SELL_SLICES;
        out.nstored = (int64_t)np;
        // what one product reads of the matrix: the values, a byte per row, the slice offsets and the table
        out.bytes = (int64_t)padded * 8 + rows + (int64_t)nslices * 4 + 4 + (int64_t)sizeof(unsigned) * PAT_MAX + 4ll * got[3];
        return LSQRHIP_OK;
    }
    (void)hipFree(out.col);
    (void)hipFree(out.val);
    out.col = nullptr;
    out.val = nullptr;
    out.sell = 3;
    out.pid = s_pid.release<unsigned char>();
    out.pdesc = s_desc.release<unsigned>();
    out.pdelta = s_delta.release<int>();
    out.pval = s_val.release<double>();
    out.npat = got[2];
    out.npat_e = got[3];
    out.nslices = (rows + 63) / 64;
    out.nblk = (out.nslices + SELL_SLICES - 1) / SELL_SLICES;
    // paired rows (pat.h): half the vector-memory requests per row.  LSQRHIP_PAT_PAIR=0 / 1.
    out.pat_pair = env_int("LSQRHIP_PAT_PAIR", 1) != 0 && out.cols >= 2;
    if (out.pat_pair) out.nblk = (out.nslices + 2 * SELL_SLICES - 1) / (2 * SELL_SLICES);
    out.nstored = 0;
    // what one product reads of the matrix: a byte per row and the table
    out.bytes = (int64_t)rows + (int64_t)sizeof(unsigned) * PAT_MAX + 12ll * got[3];
    return LSQRHIP_OK;
}

// Wide row patterns (pat.h): 257 ... 4096 distinct rows keep two bytes per row and their table in global memory.  Tried
// when the one-byte table declined for the number of patterns alone.  On success out.sell = 3 with out.pat_wide set.
static int try_pat2(hipStream_t s, Csr &out, int64_t nnz)
{
    dbg_stage(s, "try_pat2 enter");
    const int rows = out.rows;
    const int mode = env_int("LSQRHIP_PAT2", -1);
    if (out.P > 1 || nnz <= 0 || rows <= 0 || mode == 0 || env_int("LSQRHIP_PAT", -1) == 0) return LSQRHIP_OK;
    if (mode != 1 && rows < 16 * (PAT_MAX + 1)) return LSQRHIP_OK;   // (fewer than 16 rows per pattern whatever the count)
    DevScratch s_keys, s_reps, s_slot, s_len, s_ctl, s_ent, s_pid;
    HIPCHK(s_keys.alloc(sizeof(unsigned long long) * PAT2_TAB));
    HIPCHK(s_reps.alloc(sizeof(int) * PAT2_TAB));
    HIPCHK(s_len.alloc(sizeof(int) * PAT2_TAB));
    HIPCHK(s_ctl.alloc(sizeof(int) * 4));
    HIPCHK(hipMemsetAsync(s_keys.p, 0, sizeof(unsigned long long) * PAT2_TAB, s));
    HIPCHK(hipMemsetAsync(s_reps.p, 0x7f, sizeof(int) * PAT2_TAB, s));
    HIPCHK(hipMemsetAsync(s_ctl.p, 0, sizeof(int) * 4, s));
    const int g = (int)std::min<int64_t>(((int64_t)rows + 255) / 256, 2048);
    int *ctl = s_ctl.as<int>();
    hipLaunchKernelGGL(k_pat_discover, dim3(g), dim3(256), 0, s, (const int *)out.rowptr, (const int *)out.col,
                       (const double *)out.val, rows, 1, s_keys.as<unsigned long long>(), s_reps.as<int>(), ctl, PAT2_TAB,
                       PAT2_MAX);
    hipLaunchKernelGGL(k_pat2_lens, dim3(PAT2_TAB / 256), dim3(256), 0, s, (const int *)out.rowptr,
                       (const unsigned long long *)s_keys.p, (const int *)s_reps.p, PAT2_TAB, s_len.as<int>());
    HIPCHK(hipGetLastError());
    int got[4];
    HIPCHK(hipMemcpyAsync(got, ctl, sizeof(got), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int np = got[0];
    if (got[1] != 0 || np > PAT2_MAX || np <= 0) return LSQRHIP_OK;
    if (mode != 1 && (int64_t)np * 16 > rows) return LSQRHIP_OK;
    // rank the keys: pattern p = the p-th smallest key, whatever order the rows arrived in
    std::vector<unsigned long long> keys(PAT2_TAB);
    std::vector<int> len(PAT2_TAB), slot_pat(PAT2_TAB, -1);
    HIPCHK(hipMemcpyAsync(keys.data(), s_keys.p, sizeof(unsigned long long) * PAT2_TAB, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(len.data(), s_len.p, sizeof(int) * PAT2_TAB, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    std::vector<std::pair<unsigned long long, int>> order;
    order.reserve((size_t)np);
    for (int t = 0; t < PAT2_TAB; ++t)
        if (keys[(size_t)t] != 0ull) order.emplace_back(keys[(size_t)t], t);
    if ((int)order.size() != np) return LSQRHIP_OK;
    std::sort(order.begin(), order.end());
    int stride = 1;   // the longest pattern: entry k of pattern p at p * stride + k
    for (int p = 0; p < np; ++p) {
        const int t = order[(size_t)p].second, l = len[(size_t)t];
        if (l < 0 || l > PAT_MAX_LEN) return LSQRHIP_OK;
        slot_pat[(size_t)t] = p;
        stride = std::max(stride, l);
    }
    const int64_t ne = (int64_t)np * stride;
    if (ne > PAT2_MAX_E) return LSQRHIP_OK;
    HIPCHK(s_slot.alloc(sizeof(int) * PAT2_TAB));
    HIPCHK(s_ent.alloc(sizeof(PatEnt) * (size_t)ne));
    HIPCHK(s_pid.alloc(sizeof(unsigned short) * ((size_t)rows + 2)));   // (two numbers of padding: the paired-rows kernel reads two at once)
    HIPCHK(hipMemsetAsync(s_pid.as<unsigned short>() + rows, 0, 2 * sizeof(unsigned short), s));
    HIPCHK(hipMemcpyAsync(s_slot.p, slot_pat.data(), sizeof(int) * PAT2_TAB, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(s_ent.p, 0, sizeof(PatEnt) * (size_t)ne, s));
    hipLaunchKernelGGL(k_pat2_fill, dim3(PAT2_TAB / 256), dim3(256), 0, s, (const int *)out.rowptr, (const int *)out.col,
                       (const double *)out.val, (const int *)s_reps.p, (const int *)s_slot.p, PAT2_TAB, stride,
                       s_ent.as<PatEnt>());
    hipLaunchKernelGGL(k_pat2_assign, dim3(g), dim3(256), 0, s, (const int *)out.rowptr, (const int *)out.col,
                       (const double *)out.val, rows, (const unsigned long long *)s_keys.p, (const int *)s_slot.p,
                       (const PatEnt *)s_ent.p, PAT2_TAB, np, stride, s_pid.as<unsigned short>(), ctl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(got, ctl, sizeof(got), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));   // (the host vectors above are read by the copies until here)
    if (got[1] != 0) return LSQRHIP_OK;   // two different rows under one key
    (void)hipFree(out.col);
    (void)hipFree(out.val);
    out.col = nullptr;
    out.val = nullptr;
    out.sell = 3;
    out.pat_wide = true;
    out.pat_u = std::min(std::max(env_int("LSQRHIP_PAT2_U", 2), 1), 2);
    out.pat_pair = env_int("LSQRHIP_PAT_PAIR", 1) != 0 && out.cols >= 2;   // paired rows (pat.h)
    out.pat_stride = stride;
    out.pid = s_pid.release<unsigned char>();
    out.pent = s_ent.release<void>();
    out.npat = np;
    out.npat_e = (int)ne;
    out.nslices = (rows + 63) / 64;
    out.nblk = out.pat_pair ? (out.nslices + 2 * SELL_SLICES - 1) / (2 * SELL_SLICES) : (out.nslices + SELL_SLICES - 1) / SELL_SLICES;
    out.nstored = 0;
    // what one product reads of the matrix: two bytes per row and the table
    out.bytes = 2ll * rows + (int64_t)sizeof(PatEnt) * ne;
    return LSQRHIP_OK;
}

static int try_sell(hipStream_t s, Csr &out, int64_t nnz, const double *dict, int ndict, unsigned long long *stats)
{
    dbg_stage(s, "try_sell enter");
    const int rows = out.rows;
    // LSQRHIP_SELL: 0 never | 1 whenever the row shape qualifies | unset: also require local
    // columns (every slice spans < 65536 columns) -- with scattered columns the layout buys
    // nothing (x gathers dominate; measured 5-10 % slower than row windows)
    const int mode = env_int("LSQRHIP_SELL", -1);
    if (out.P > 1 || nnz <= 0 || rows <= 0 || mode == 0) return LSQRHIP_OK;
    if (nnz > 24 * (int64_t)rows) return LSQRHIP_OK;
    const int nslices = (rows + 63) / 64;
    const unsigned gr = (unsigned)(((int64_t)nslices * 64 + 255) / 256);
    DevScratch s_off, s_cb, s_col, s_val, s_len;  // released into `out` only on success
    HIPCHK(s_off.alloc(sizeof(unsigned) * ((size_t)nslices + 1)));
    unsigned *soff = s_off.as<unsigned>();
    HIPCHK(hipMemsetAsync(stats, 0, 4 * sizeof(unsigned long long), s));
    HIPCHK(hipMemsetAsync(soff + nslices, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_sell_width, dim3(gr), dim3(256), 0, s, (const int *)out.rowptr, rows, nslices, soff, stats);
    HIPCHK(hipGetLastError());
    unsigned long long st[4];
    HIPCHK(hipMemcpyAsync(st, stats, sizeof(st), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const unsigned long long padded = st[0];
    if (st[1] > (unsigned long long)SELL_MAX_W || padded > (unsigned long long)(nnz + nnz / 8 + 4096) ||
        padded >= (1ull << 31)) {
        return LSQRHIP_OK;
    }
    HIPCHK(s_cb.alloc(sizeof(int) * (size_t)nslices));
    int *cbaseS = s_cb.as<int>();
    hipLaunchKernelGGL(k_sell_colspan, dim3(gr), dim3(256), 0, s, (const int *)out.rowptr, (const int *)out.col, rows,
                       nslices, cbaseS, stats);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(st, stats, sizeof(st), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (st[2] != 0 && mode != 1) return LSQRHIP_OK;
    const bool c16 = st[2] == 0 && env_int("LSQRHIP_COL16", 1) != 0;
    const bool v8 = ndict > 0;
    // 3-byte nonzeros: packed 16-byte records when they cost <= 10 % more than the slices (sell.h)
    const int pmode = env_int("LSQRHIP_SELLP", -1);
    const unsigned long long nrec = st[3];
    const bool packed = c16 && v8 && pmode != 0 && nrec < (1ull << 31) &&
                        (pmode == 1 || 10 * 16 * nrec <= 11 * (3 * padded + (unsigned long long)rows));
    if (packed) {
        hipLaunchKernelGGL(k_sellp_chunks, dim3((unsigned)((nslices + 255) / 256)), dim3(256), 0, s, soff, nslices);
        launch_scan_small(s, soff, (int64_t)nslices + 1);
        HIPCHK(hipGetLastError());
        DevScratch s_rec;
        HIPCHK(s_rec.alloc(sizeof(uint4) * (size_t)std::max<unsigned long long>(nrec, 1)));
        hipLaunchKernelGGL(k_sellp_fill, dim3(gr), dim3(256), 0, s, (const int *)out.rowptr, (const int *)out.col,
                           (const double *)out.val, (const unsigned *)soff, (const int *)cbaseS,
                           (const unsigned long long *)dict, ndict, rows, nslices, s_rec.as<uint4>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        (void)hipFree(out.col);
        (void)hipFree(out.val);
        out.col = nullptr;
        out.val = nullptr;
        out.sell = 2;
        out.soff = s_off.release<unsigned>();
        out.srec = s_rec.release<uint4>();
        out.cbaseS = s_cb.release<int>();
        out.nslices = nslices;
        out.sell_c16 = true;
        out.sell_v8 = true;
        out.dict = dict;
        out.nblk = (nslices + SELL_SLICES - 1) / SELL_SLICES;
        out.bytes = (int64_t)nrec * 16 + (int64_t)nslices * 8 + 4;
        return LSQRHIP_OK;
    }
    launch_scan_small(s, soff, (int64_t)nslices + 1);
    HIPCHK(hipGetLastError());
    const size_t np = (size_t)std::max<unsigned long long>(padded, 1);
    HIPCHK(s_col.alloc(np * (c16 ? 2 : 4)));
    HIPCHK(s_val.alloc(np * (v8 ? 1 : 8)));
    HIPCHK(s_len.alloc((size_t)rows));
    void *scol = s_col.p, *sval = s_val.p;
    unsigned char *rlen = s_len.as<unsigned char>();
    const unsigned long long *db = (const unsigned long long *)dict;
#define SELL_FILL(C16, V8)                                                                                          \
    hipLaunchKernelGGL((k_sell_fill<C16, V8>), dim3(gr), dim3(256), 0, s, (const int *)out.rowptr,                  \
                       (const int *)out.col, (const double *)out.val, (const unsigned *)soff, (const int *)cbaseS, \
                       db, ndict, rows, nslices, scol, sval, rlen)
    if (c16 && v8) SELL_FILL(true, true);
    else if (c16) SELL_FILL(true, false);
    else if (v8) SELL_FILL(false, true);
    else SELL_FILL(false, false);
#undef SELL_FILL
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    (void)hipFree(out.col);
    (void)hipFree(out.val);
    out.col = nullptr;
    out.val = nullptr;
    out.sell = 1;
    out.nstored = (int64_t)np;
    out.soff = s_off.release<unsigned>();
    out.scol = s_col.release<void>();
    out.sval = s_val.release<void>();
    out.cbaseS = s_cb.release<int>();
    out.rlen = s_len.release<unsigned char>();
    out.nslices = nslices;
    out.sell_c16 = c16;
    out.sell_v8 = v8;
    out.dict = dict;
    out.nblk = (nslices + SELL_SLICES - 1) / SELL_SLICES;
    // what one product reads of the matrix (the row pointers stay allocated but are not read)
    out.bytes = (int64_t)padded * ((c16 ? 2 : 4) + (v8 ? 1 : 8)) + (int64_t)nslices * 8 + 4 + rows;
    return LSQRHIP_OK;
}

// in-place exclusive scan of a[0..L); `sums` is scratch of L / SCAN_CHUNK + 1 words
static void launch_scan(hipStream_t s, unsigned *a, int64_t L, unsigned *sums)
{
    if (L <= 4 * SCAN_CHUNK) {
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, a, L);
        return;
    }
    const int64_t nc = (L + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nc), dim3(256), 0, s, (const unsigned *)a, L, sums);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, s, sums, nc);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nc), dim3(256), 0, s, a, L, (const unsigned *)sums);
}

template <typename OffT>
static int build_csr_T(hipStream_t s, const int *d_keys, const int *d_other, const double *d_a, int64_t nnz,
                       int rows, int cols, int bad_code, int bad_code_other, int panels, int pw, int xlds,
                       unsigned long long *bufA, unsigned long long *bufB, unsigned *hist, int *d_flags,
                       const double *dict, int ndict, Csr &out)
{
    out.rows = rows;
    out.cols = cols;
    out.P = panels > 1 ? panels : 1;
    out.pw = out.P > 1 ? pw : cols;
    out.xlds = out.P > 1 ? xlds : 0;
    if (out.xlds && env_int("LSQRHIP_XLDS_WG", 1024) == 1024) out.xlds = 2;
    out.rows_v = (int64_t)out.P * rows;
    const int rows_v = (int)out.rows_v;  // < 2^31, checked by the caller
    HIPCHK(hipMalloc(&out.rowptr, sizeof(OffT) * ((size_t)rows_v + 1)));
    HIPCHK(hipMalloc((void **)&out.col, sizeof(int) * (size_t)std::max<int64_t>(nnz, 1)));
    HIPCHK(hipMalloc((void **)&out.val, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1)));
    out.bytes = (int64_t)sizeof(OffT) * (rows_v + 1) + 12 * nnz;
    out.nstored = nnz;
    probe_mem();

    HIPCHK(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
    const int g = (int)std::min<int64_t>(std::max<int64_t>((nnz + 255) / 256, 1), 4096);
    int flags[4] = {0, 0, 0, 0};
    unsigned long long *sorted = bufA;
    if (nnz > 0) {
        if (out.P > 1) {  // key = panel(col) * rows + row: never pre-sorted
            hipLaunchKernelGGL(k_pack_keys_panel, dim3(g), dim3(256), 0, s, d_keys, d_other, nnz, rows, cols, out.pw,
                               bufA, d_flags);
            flags[1] = 1;
        } else {
            hipLaunchKernelGGL(k_pack_keys, dim3(g), dim3(256), 0, s, d_keys, nnz, rows, bufA, d_flags);
        }
        HIPCHK(hipGetLastError());
        int got[4];
        HIPCHK(hipMemcpyAsync(got, d_flags, sizeof(got), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (got[0]) return fail(bad_code, lsqrhip_error_string(bad_code));
        if (got[2]) return fail(bad_code_other, lsqrhip_error_string(bad_code_other));
        flags[1] |= got[1];
        if (flags[1]) {  // not sorted by key: stable LSD radix sort on the key bits
            const int64_t nb = (nnz + RS_TILE - 1) / RS_TILE;
            const int nbits = bits_for(rows_v);
            unsigned long long *in = bufA, *outb = bufB;
            for (int shift = 32; shift < 32 + nbits; shift += 8) {
                hipLaunchKernelGGL(k_radix_hist, dim3((unsigned)nb), dim3(RS_BLOCK), 0, s, in, nnz, shift, nb, hist);
                launch_scan(s, hist, (int64_t)256 * nb, hist + (size_t)256 * nb);
                hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)nb), dim3(RS_BLOCK), 0, s, in, outb, nnz, shift, nb, hist);
                HIPCHK(hipGetLastError());
                std::swap(in, outb);
            }
            sorted = in;
        }
        hipLaunchKernelGGL(k_csr_gather, dim3(g), dim3(256), 0, s, sorted, nnz, d_other, d_a, out.col, out.val);
    }
    hipLaunchKernelGGL(k_rowptr_from_sorted<OffT>, dim3(g), dim3(256), 0, s, sorted, nnz, rows_v, (OffT *)out.rowptr);
    HIPCHK(hipGetLastError());

    dbg_stage(s, "csr built");
    // short, even rows: sliced-ELL layout instead of row windows (sell.h)
    if (std::is_same<OffT, int>::value) {
        bool too_many = false;
        int rcs = try_pat(s, out, nnz, true, (unsigned long long *)hist, &too_many);   // rows that repeat: one byte per row (pat.h)
        // ... more than 256 of them: two bytes per row, the table through L2 (pat.h "wide")
        if (rcs == LSQRHIP_OK && !out.sell && too_many) rcs = try_pat2(s, out, nnz);
        // ... rows whose column structure repeats, without a value dictionary: no column indices (pat.h)
        if (rcs == LSQRHIP_OK && !out.sell && ndict == 0) rcs = try_pat(s, out, nnz, false, (unsigned long long *)hist);
        if (rcs == LSQRHIP_OK && !out.sell) rcs = try_sell(s, out, nnz, dict, ndict, (unsigned long long *)hist);
        if (rcs != LSQRHIP_OK) return rcs;
    }
    dbg_stage(s, "short-row layouts tried");
    if (out.sell) {
        // 6 workgroups per CU: measured best at every size (config 2: 25.0 vs 26.1 us per iteration
        // with 8 per CU; fewer partial sums to re-read, still enough waves for the streams)
        // row patterns (pat.h): 4 per CU, two slices per wave and trip -- config 2: 46.5k iterations/s, against 44.9k /
        // 43.3k with 896 / 1152 workgroups and 45.4k with 1536 and one slice per trip (profiles/r03/config2_patterns.txt)
        const int cap = out.sell == 3 ? std::min(std::max(env_int("LSQRHIP_PAT_GRID", PAT_MAX_GRID), 8), SPMV_MAX_GRID)
                                      : std::min(std::max(env_int("LSQRHIP_SELL_GRID", SELL_MAX_GRID), 8), SPMV_MAX_GRID);
        int64_t grid = std::min<int64_t>(out.nblk, cap);
        if (grid >= 8) grid &= ~(int64_t)7;
        out.grid = (int)std::max<int64_t>(grid, 1);
        out.out_grid = out.grid;
        return LSQRHIP_OK;
    }

    // row blocks ("row windows", spmv.h) over the (virtual) rows
    const int winC = out.xlds == 2 ? XLW_C : SPMV_C;  // wave-sized windows for xl.h
    out.nblk = std::max<int64_t>((nnz + rows_v + winC - 1) / winC, 1);
    HIPCHK(hipMalloc((void **)&out.rb, sizeof(int) * (size_t)(out.nblk + 1)));
    out.bytes += (int64_t)sizeof(int) * (out.nblk + 1);
    const int64_t nt = out.nblk + 1;
    hipLaunchKernelGGL(k_row_blocks<OffT>, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s,
                       (const OffT *)out.rowptr, rows_v, out.nblk, out.rb, winC);
    HIPCHK(hipMalloc((void **)&out.blk, sizeof(RowBlock) * (size_t)out.nblk));
    out.bytes += (int64_t)sizeof(RowBlock) * out.nblk;
    hipLaunchKernelGGL(k_block_desc<OffT>, dim3((unsigned)((out.nblk + 255) / 256)), dim3(256), 0, s,
                       (const OffT *)out.rowptr, (const int *)out.rb, out.nblk, out.blk);
    HIPCHK(hipGetLastError());
    if (out.P > 1 && out.xlds != 2 && env_int("LSQRHIP_SKEW", 1) != 0) {  // skewed windows (spmv.h phase 2b)
        HIPCHK(hipMalloc((void **)&out.skew, (size_t)out.nblk));
        HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_block_skew<OffT>, dim3((unsigned)((out.nblk + 255) / 256)), dim3(256), 0, s,
                           (const OffT *)out.rowptr, (const RowBlock *)out.blk, out.nblk, out.skew, d_flags);
        int any = 0;
        HIPCHK(hipMemcpyAsync(&any, d_flags, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (!any) {  // nothing to set aside anywhere: the kernel skips the lookup
            (void)hipFree(out.skew);
            out.skew = nullptr;
        } else {
            out.bytes += out.nblk;
        }
    }
    if (out.xlds == 2) {  // xl.h: the panel of every trip of XLW_WAVES windows
        const int64_t ngrp = (out.nblk + XLW_WAVES - 1) / XLW_WAVES;
        HIPCHK(hipMalloc((void **)&out.gpid, sizeof(int) * (size_t)ngrp));
        hipLaunchKernelGGL(k_xl_group_panel, dim3((unsigned)((ngrp + 255) / 256)), dim3(256), 0, s,
                           (const RowBlock *)out.blk, out.nblk, rows, ngrp, out.gpid);
        HIPCHK(hipGetLastError());
    }
    // wave-window LDS panels (xl.h): 16-bit columns relative to the window's own panel, always possible
    if (out.xlds == 2 && nnz > 0 && env_int("LSQRHIP_COL16", 1) != 0) {
        HIPCHK(hipMalloc((void **)&out.col16, sizeof(unsigned short) * (size_t)nnz));
        const unsigned gb = (unsigned)std::min<int64_t>((out.nblk + 3) / 4, 65535);
        hipLaunchKernelGGL(k_xl_col16, dim3(gb), dim3(256), 0, s, (const RowBlock *)out.blk, out.nblk,
                           (const int *)out.col, rows, out.pw, out.col16);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        (void)hipFree(out.col);
        out.col = nullptr;
        out.bytes -= 2 * nnz;
        // ... and 16-bit row bounds relative to the same windows; the row pointers are no longer needed
        HIPCHK(hipMalloc((void **)&out.rel16, sizeof(unsigned short) * ((size_t)rows_v + 2)));
        HIPCHK(hipMemsetAsync(out.rel16, 0, sizeof(unsigned short) * ((size_t)rows_v + 2), s));
        hipLaunchKernelGGL(k_xl_rel16<OffT>, dim3(gb), dim3(256), 0, s, (const RowBlock *)out.blk, out.nblk,
                           (const OffT *)out.rowptr, out.rel16);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        (void)hipFree(out.rowptr);
        out.rowptr = nullptr;
        out.bytes -= (int64_t)(sizeof(OffT) - 2) * rows_v;
    }
    // 16-bit block-relative columns when every row block is narrower than 65536 columns
    // (LSQRHIP_COL16=0 keeps 32-bit indices)
    if (out.P <= 1 && nnz > 0 && env_int("LSQRHIP_COL16", 1) != 0) {
        HIPCHK(hipMalloc((void **)&out.cbase, sizeof(int) * (size_t)out.nblk));
        HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), s));
        const unsigned gb = (unsigned)std::min<int64_t>(out.nblk, 65535);
        hipLaunchKernelGGL(k_block_colspan, dim3(gb), dim3(256), 0, s, (const RowBlock *)out.blk, out.nblk,
                           (const int *)out.col, out.cbase, d_flags);
        int wide = 0;
        HIPCHK(hipMemcpyAsync(&wide, d_flags, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (!wide) {
            HIPCHK(hipMalloc((void **)&out.col16, sizeof(unsigned short) * (size_t)nnz));
            hipLaunchKernelGGL(k_col_to16, dim3(gb), dim3(256), 0, s, (const RowBlock *)out.blk, out.nblk,
                               (const int *)out.col, (const int *)out.cbase, out.col16);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s));
            (void)hipFree(out.col);
            out.col = nullptr;
            out.bytes += (int64_t)sizeof(int) * out.nblk - 2 * nnz;
        } else {
            (void)hipFree(out.cbase);
            out.cbase = nullptr;
        }
    }
    // one-byte value codes when the matrix has a dictionary (valdict.h)
    if (ndict > 0 && nnz > 0) {
        HIPCHK(hipMalloc((void **)&out.val8, (size_t)nnz));
        hipLaunchKernelGGL(k_dict_encode, dim3(g), dim3(256), 0, s, (const double *)out.val, nnz,
                           (const unsigned long long *)dict, ndict, out.val8);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        (void)hipFree(out.val);
        out.val = nullptr;
        out.dict = dict;
        out.bytes -= 7 * nnz;
    }
    int64_t grid = std::min<int64_t>(out.nblk, std::min(std::max(env_int("LSQRHIP_SPMV_GRID", SPMV_MAX_GRID), 8), SPMV_MAX_GRID));
    if (grid >= 8) grid &= ~(int64_t)7;  // multiple of 8: XCD-aware mapping (common.h)
    out.grid = (int)std::max<int64_t>(grid, 1);
    out.out_grid = out.P > 1 ? vec_grid(2 * (int64_t)rows) : out.grid;
    {   // xl.h: one 1024-thread workgroup per CU, trips of XLW_WAVES windows
        int64_t xg = std::min<int64_t>((out.nblk + XLW_WAVES - 1) / XLW_WAVES, 256);
        if (xg >= 8) xg &= ~(int64_t)7;
        out.xgrid = (int)std::max<int64_t>(xg, 1);
    }
    HIPCHK(hipStreamSynchronize(s));
    return LSQRHIP_OK;
}

// Stable LSD radix sort of 64-bit words on their bits [32, 32 + nbits) (csr_build.h); returns the
// buffer that holds the sorted words (`in` or `tmp`).
static unsigned long long *radix_sort_words(hipStream_t s, unsigned long long *in, unsigned long long *tmp, int64_t nnz,
                                            int nbits, unsigned *hist)
{
    const int64_t nb = (nnz + RS_TILE - 1) / RS_TILE;
    for (int shift = 32; shift < 32 + nbits; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3((unsigned)nb), dim3(RS_BLOCK), 0, s, in, nnz, shift, nb, hist);
        launch_scan(s, hist, (int64_t)256 * nb, hist + (size_t)256 * nb);
        hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)nb), dim3(RS_BLOCK), 0, s, in, tmp, nnz, shift, nb, hist);
        std::swap(in, tmp);
    }
    return in;
}

// The parts an n-vector is exchanged in by the sharded engine with LSQRHIP_SHARD_OVERLAP=1: slice q of P (columns
// [q c, (q + 1) c), c = ceil(n / P)) in G parts of cg = ceil(c / G) columns.  Part (q, k) = [lo, hi), clipped to n.
static void shard_part(int64_t n, int P, int G, int q, int k, int64_t *lo, int64_t *hi)
{
    const int64_t c = (n + P - 1) / P, cg = (c + G - 1) / G;
    *lo = std::min<int64_t>(n, (int64_t)q * c + std::min<int64_t>((int64_t)k * cg, c));
    *hi = std::min<int64_t>(n, (int64_t)q * c + std::min<int64_t>((int64_t)(k + 1) * cg, c));
}
// What the overlap asks of a layout: `stripes` -- the chunks of every row block are formed per part of the gathered
// vector (mode 1: v arrives part by part) --, `segments` -- the row blocks are cut per part of the OUTPUT vector and
// launched part-major (mode 2: T leaves part by part).  P = 0: neither.
struct CsbPlan {
    int P = 0, G = 2;
    bool stripes = false, segments = false;
};

// Column-swept row blocks (csb.h) of the product whose rows are `rowk` and whose gathered vector is
// indexed by `colk`.  On success out.csb = 1; when a chunk would span 2^18 columns or more (an almost
// empty row block) `out` is left untouched and the caller builds the panel layout instead.
//   LSQRHIP_CSB_R  rows per block (test hook; default: as many as the LDS holds, cut so that the
//                  blocks divide evenly among the 256 workgroups)
static int build_csb(hipStream_t s, const int *rowk, const int *colk, const double *d_a, int64_t nnz, int rows,
                     int cols, bool f32, int bad_code, int bad_code_other, DevScratch &sbufA, DevScratch &sbufB,
                     unsigned *hist, int *d_flags, Csr &out, const CsbPlan &plan = CsbPlan())
{
    if (rows <= 0 || cols <= 0) return LSQRHIP_OK;
    // (the two sort buffers are the caller's, 8 bytes per nonzero each: allocated here when a previous build released
    //  them, released again while the layout's own arrays are filled -- see below -- and NOT handed back: the caller
    //  allocates them again only if something still needs them, so that they never sit beside two finished layouts)
    const size_t sort_bytes = sizeof(unsigned long long) * (size_t)std::max<int64_t>(nnz, 1);
    if (!sbufA.p) HIPCHK(sbufA.alloc(sort_bytes));
    if (!sbufB.p) HIPCHK(sbufB.alloc(sort_bytes));
    unsigned long long *bufA = sbufA.as<unsigned long long>(), *bufB = sbufB.as<unsigned long long>();
    const int g = (int)std::min<int64_t>(std::max<int64_t>((nnz + 255) / 256, 1), 4096);
    HIPCHK(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
    int got[4] = {0, 0, 0, 0};
    DevScratch s_pos, s_cnt, s_rbs, s_nrm, s_emax, s_rexp;
    HIPCHK(s_pos.alloc(sizeof(unsigned) * (size_t)std::max<int64_t>(nnz, 1)));
    HIPCHK(s_cnt.alloc(sizeof(int) * ((size_t)rows + 1)));
    HIPCHK(hipMemsetAsync(s_cnt.p, 0, sizeof(int) * ((size_t)rows + 1), s));
    // per row: its largest exponent, its 1-norm as an integer sum relative to that (csb.h k_csb_pos), and from the
    // two e1_i with 2^e1_i > sum_j |a_ij| -- the power of two the row's stored values are divided by
    HIPCHK(s_nrm.alloc(sizeof(unsigned long long) * (size_t)rows));
    HIPCHK(hipMemsetAsync(s_nrm.p, 0, sizeof(unsigned long long) * (size_t)rows, s));
    HIPCHK(s_emax.alloc(sizeof(int) * (size_t)rows));
    HIPCHK(s_rexp.alloc(sizeof(short) * (size_t)rows));
    unsigned long long *sorted1 = bufA;
    int maxrow = 0;
    const dim3 gr((unsigned)std::min<int64_t>(((int64_t)rows + 255) / 256, 2048));
    if (nnz > 0) {
        hipLaunchKernelGGL(k_csb_pack_col, dim3(g), dim3(256), 0, s, rowk, colk, nnz, rows, cols, bufA, d_flags);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(got, d_flags, sizeof(got), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (got[0]) return fail(bad_code, lsqrhip_error_string(bad_code));
        if (got[2]) return fail(bad_code_other, lsqrhip_error_string(bad_code_other));
        if (got[1]) sorted1 = radix_sort_words(s, bufA, bufB, nnz, bits_for(cols), hist);
        hipLaunchKernelGGL(k_fill_int, gr, dim3(256), 0, s, s_emax.as<int>(), (int64_t)rows, CSB_NO_EXP);
        HIPCHK(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(k_csb_rowemax, dim3(g), dim3(256), 0, s, rowk, d_a, nnz, s_emax.as<int>(), d_flags);
        unsigned long long *n1 = s_nrm.as<unsigned long long>();
        hipLaunchKernelGGL(k_csb_pos, dim3(g), dim3(256), 0, s, (const unsigned long long *)sorted1, nnz, rowk, d_a,
                           (const int *)s_emax.as<int>(), s_pos.as<unsigned>(), s_cnt.as<int>(), n1);
        hipLaunchKernelGGL(k_csb_rexp, dim3((unsigned)(((int64_t)rows + 255) / 256)), dim3(256), 0, s,
                           (const unsigned long long *)n1, (const int *)s_emax.as<int>(), rows, s_rexp.as<short>(), d_flags);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(got, d_flags, sizeof(got), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        // a value that is not finite, or a row norm beyond 2^+-900: not this layout (csb.h)
        if (got[0] || got[1]) return LSQRHIP_OK;
        // the longest row (how even the rows are)
        HIPCHK(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(k_csb_maxint, gr, dim3(256), 0, s, (const int *)s_cnt.as<int>(), (int64_t)rows, d_flags);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(got, d_flags, sizeof(got), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        maxrow = got[0];
    } else {
        HIPCHK(hipMemsetAsync(s_rexp.p, 0, sizeof(short) * (size_t)rows, s));
    }
    s_emax.free_now();
    s_nrm.free_now();
    // near-uniform rows (the longest <= 512): blocks cut by nonzeros stay close to the mean row count
    const bool even_rows = maxrow <= 512;
    const int rmax = CSB_RMAX;
    // Blocks of at most rmax rows, cut so that every block holds about the same number of NONZEROS and so that
    // every launch ("round") of 256 workgroups has one whole block, or one column split of a block, per CU:
    //   * enough rows for 256 full blocks: 256 k blocks, k the smallest for which the cut stays within them
    //     (blocks are cut by nonzeros, so where rows are short a block takes more rows than the mean; when that
    //     runs into what the LDS holds the surplus moves to later blocks -- and if it does not fit, k grows);
    //   * fewer rows: S workgroups share a block (column splits, csb.h) so that blocks stay tall -- what counts
    //     is R d / n, the nonzeros a block holds per column of x.
    //   LSQRHIP_CSB_S  splits per block (test hook; default: as many as keep blocks within rmax rows, at most 8)
    //   LSQRHIP_CSB_R  rows per block, exactly (test hook)
    std::vector<int> cnt((size_t)rows);
    HIPCHK(hipMemcpy(cnt.data(), s_cnt.p, sizeof(int) * (size_t)rows, hipMemcpyDeviceToHost));
    // nb blocks by cumulative nonzeros (block k ends where the running count reaches k / nb of the total), none
    // longer than `cap` rows; or, by_rows, blocks of exactly `cap` rows
    auto cut = [&](int nb, int cap, bool by_rows) {
        std::vector<int> rs;
        rs.push_back(0);
        const double target = (double)nnz / (double)std::max(nb, 1);
        double acc = 0.0;
        int inblk = 0;
        for (int r = 0; r < rows; ++r) {
            acc += cnt[(size_t)r];
            ++inblk;
            const bool full = inblk >= cap;
            const bool enough = !by_rows && acc >= target * (double)rs.size();
            if ((full || enough) && r + 1 < rows) {
                rs.push_back(r + 1);
                inblk = 0;
            }
        }
        rs.push_back(rows);
        return rs;
    };
    const int rfill = even_rows ? (int)(0.97 * rmax) : (int)(0.9 * rmax);
    // rounds of 256 / S blocks: the fewest for which the cut by nonzeros stays within them
    auto cut_rounds = [&](int S_, std::vector<int> &rs) {
        const int per_round = std::max(1, CSB_GRID / S_);
        const int kmin = (int)std::max<int64_t>(1, ((int64_t)rows + (int64_t)per_round * rmax - 1) / ((int64_t)per_round * rmax));
        for (int k = kmin;; ++k) {
            // (small systems: a few whole blocks of >= 512 rows rather than 256 slivers)
            const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)per_round * k, ((int64_t)rows + 511) / 512));
            rs = cut(nb, rmax, false);
            if ((int)rs.size() - 1 <= per_round * k || k >= kmin + 4) return;
        }
    };
    int S = env_int("LSQRHIP_CSB_S", 0);
    std::vector<int> rstart;
    const int r_forced = env_int("LSQRHIP_CSB_R", 0);
    if (S > 0) {
        S = std::min(S, 8);
    } else {
        // few rows: as many splits as keep one round of blocks tall
        S = (int)std::min<int64_t>(8, std::max<int64_t>(1, (int64_t)rfill * CSB_GRID / std::max(rows, 1)));
        if ((int64_t)rows * S < (int64_t)CSB_GRID * 512) S = 1;   // small systems: not worth a second launch
        // Many rows (a round of full blocks or more): NO splits since round 5.  Rounds 3 and 4 gave such matrices 2 or 4
        // splits when x lay far beyond L2 -- every XCD then sweeps a half / a quarter of x per launch and fewer of its
        // gathers miss L2 (config 4: PMC fetch 25 GB -> 16.4 GB, 3.9 -> 3.45 ms with the free-running sweep).  Under the
        // lock-step sweep the misses cost less than the splits' partial sums and the combine launch do
        // (profiles/r05/lockstep_splits_by_shape.txt, one process per shape, ms mode 1 / mode 2):
        //   config 4 (10M x 10M)                 S = 1 2.82 / 2.82   2: 2.98 / 2.93   4: 3.09 / 3.13   8: 3.37 / 3.32
        //   one rank of two's block (5M x 10M)   S = 1 1.41 / 1.48   2: 1.50 / 1.59   4: 1.58 / 1.69
        // Blocks of fewer rows than a round (one rank of four, of eight) keep the splits that fill the chip, above.
        if (env_int("LSQRHIP_CSB_SPLIT_RULE", 5) == 4 && S == 1 && 2 * (int64_t)rows > (int64_t)CSB_GRID * rmax &&
            (int64_t)cols * 8 > (32ll << 20))   // (round 4's rule, for the A/B)
            S = ((int64_t)rows > (int64_t)CSB_GRID * rmax && (int64_t)cols * 8 >= (64ll << 20)) ? 4 : 2;
    }
    std::vector<int> border, phase_pos;   // segments: launch order of the blocks, where its phases begin
    const bool segments = plan.segments && plan.P > 1 && plan.G > 1 && nnz > 0 && r_forced <= 0 && S == 1;
    if (segments) {
        // Row blocks cut per part (q, k) of the rows, by nonzeros inside each part, and launched part-major: all
        // parts k = 0 first -- one round (or a few) of 256 blocks after which T's first parts are complete -- then
        // k = 1, ...  Blocks per part: an equal share of the rounds a phase gets.
        const int P = plan.P, G = plan.G;
        const int kmin = (int)std::max<int64_t>(1, ((int64_t)rows + (int64_t)CSB_GRID * rmax - 1) / ((int64_t)CSB_GRID * rmax));
        const int rpp = std::max(1, (kmin + G - 1) / G);                   // rounds per phase
        const int nbs = std::max(1, CSB_GRID * rpp / P);                   // blocks per part
        std::vector<std::vector<int>> first((size_t)P * G);                // first rows of the blocks of every part
        rstart.clear();
        std::vector<int> part_first_block((size_t)P * G + 1, 0);
        for (int q = 0; q < P; ++q)
            for (int k = 0; k < G; ++k) {
                int64_t lo, hi;
                shard_part(rows, P, G, q, k, &lo, &hi);
                part_first_block[(size_t)q * G + k] = (int)rstart.size();
                if (hi <= lo) continue;
                double tot = 0.0;
                for (int64_t r = lo; r < hi; ++r) tot += cnt[(size_t)r];
                const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(nbs, (hi - lo + 511) / 512));
                const double target = tot / nb;
                double acc = 0.0;
                int inblk = 0, made = 1;
                rstart.push_back((int)lo);
                for (int64_t r = lo; r < hi; ++r) {
                    acc += cnt[(size_t)r];
                    ++inblk;
                    if ((inblk >= rmax || acc >= target * made) && r + 1 < hi) {
                        rstart.push_back((int)(r + 1));
                        inblk = 0;
                        ++made;
                    }
                }
            }
        part_first_block[(size_t)P * G] = (int)rstart.size();
        rstart.push_back(rows);
        for (int k = 0; k < G; ++k) {
            phase_pos.push_back((int)border.size());
            for (int q = 0; q < P; ++q)
                for (int b = part_first_block[(size_t)q * G + k]; b < part_first_block[(size_t)q * G + k + 1]; ++b) border.push_back(b);
        }
        phase_pos.push_back((int)border.size());
    } else if (r_forced > 0 || nnz <= 0) {
        const int R = std::min(std::max(r_forced > 0 ? r_forced : rmax, 1), rmax);
        rstart = cut(0, R, true);
    } else {
        cut_rounds(S, rstart);
    }
    cnt.clear();
    cnt.shrink_to_fit();
    const int nrb = (int)rstart.size() - 1;
    // Column stripes (mode 1 of a rank of the overlapping engine): the chunks of a block are formed per part (q, k) of
    // the gathered vector; split sp = k * J + j of a block sweeps part k of the slices j, j + J, ...  J sub-splits so
    // that one phase (all blocks x J splits) fills the chip.
    int NS = 1, G = 1, J = 1;
    const int S_plain = S;   // the splits this matrix has without the plan
    std::vector<int> scut;
    if (plan.stripes && plan.P > 1 && plan.G > 1 && nnz > 0 && env_int("LSQRHIP_CSB_S", 0) <= 0) {
        G = plan.G;
        J = std::max(1, std::min(S, 8 / G));   // (S so far: the splits that keep one round of blocks tall)
        while (J > 1 && plan.P % J != 0) --J;  // every sub-split the same number of slices
        S = G * J;
        NS = plan.P * G;
        for (int q = 0; q < plan.P; ++q)
            for (int k = 0; k < G; ++k) {
                int64_t lo, hi;
                shard_part(cols, plan.P, G, q, k, &lo, &hi);
                scut.push_back((int)lo);
            }
        scut.push_back(cols);
    }
    if (nrb > SPMV_MAX_GRID) return LSQRHIP_OK;  // one partial of sum(y^2) per block
    DevScratch s_rst;
    HIPCHK(s_rst.alloc(sizeof(int) * rstart.size()));
    HIPCHK(hipMemcpyAsync(s_rst.p, rstart.data(), sizeof(int) * rstart.size(), hipMemcpyHostToDevice, s));
    if ((int64_t)nrb * NS >= (1ll << 30)) return LSQRHIP_OK;
    const int ngrp = nrb * NS;   // groups = (block, stripe): the units chunks are formed in
    HIPCHK(s_rbs.alloc(sizeof(long long) * ((size_t)ngrp + 1)));
    DevScratch s_scut;
    if (NS > 1) {
        HIPCHK(s_scut.alloc(sizeof(int) * scut.size()));
        HIPCHK(hipMemcpyAsync(s_scut.p, scut.data(), sizeof(int) * scut.size(), hipMemcpyHostToDevice, s));
    }
    unsigned long long *sorted2 = bufA;
    if (nnz > 0) {
        hipLaunchKernelGGL(k_csb_pack_rb, dim3(g), dim3(256), 0, s, rowk, colk, (const unsigned *)s_pos.as<unsigned>(), nnz,
                           (const int *)s_rst.as<int>(), nrb, (const int *)s_scut.as<int>(), NS, bufA);
        HIPCHK(hipGetLastError());
        sorted2 = ngrp > 1 ? radix_sort_words(s, bufA, bufB, nnz, bits_for(ngrp), hist) : bufA;
    }
    hipLaunchKernelGGL(k_rowptr_from_sorted<long long>, dim3(g), dim3(256), 0, s, (const unsigned long long *)sorted2, nnz,
                       ngrp, s_rbs.as<long long>());
    HIPCHK(hipGetLastError());
    std::vector<long long> rbs((size_t)ngrp + 1), gptr((size_t)ngrp + 1), cptr((size_t)nrb + 1);
    HIPCHK(hipMemcpyAsync(rbs.data(), s_rbs.p, sizeof(long long) * rbs.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    gptr[0] = 0;
    for (int gi = 0; gi < ngrp; ++gi) gptr[gi + 1] = gptr[gi] + (rbs[gi + 1] - rbs[gi] + CSB_CHUNK - 1) / CSB_CHUNK;
    for (int b = 0; b <= nrb; ++b) cptr[b] = gptr[(size_t)b * NS];
    const long long nchunks = cptr[nrb];
    if (nchunks >= (1ll << 31)) return LSQRHIP_OK;
    // The sorts are done: what the fill needs of them is, per element of the layout, the COO position it comes from --
    // 4 bytes, composed here from the block-order words (8) and the column-order positions (4).  The sort buffers and
    // the positions are released before the layout's 12 bytes per nonzero are allocated: the build peaks at the
    // triplets + 8 + 8 + 4 + 4 bytes per nonzero while it sorts and at the triplets + 4 + 12 while it fills, not at
    // their sum, and the sort buffers never sit beside two finished layouts (the literal config 3, 4e9 nonzeros: the
    // peak beyond the 64 GB of triplets is A + 8 + 8 + 4 bytes per nonzero of the second build = 128 GB for a 96 GB result).
    HIPCHK(hipStreamSynchronize(s));
    if (sorted2 == bufA) sbufB.free_now();   // (the sort's other buffer is done with)
    else sbufA.free_now();
    DevScratch s_perm;
    HIPCHK(s_perm.alloc(sizeof(unsigned) * (size_t)std::max<int64_t>(nnz, 1)));
    probe_mem();
    if (nnz > 0)
        hipLaunchKernelGGL(k_csb_compose, dim3(g), dim3(256), 0, s, (const unsigned long long *)sorted2,
                           (const unsigned *)s_pos.as<unsigned>(), nnz, s_perm.as<unsigned>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    s_pos.free_now();
    sbufA.free_now();
    sbufB.free_now();
    bufA = bufB = sorted2 = nullptr;
    DevScratch s_val, s_idx, s_cb, s_cptr;
    const size_t ne = (size_t)std::max<long long>(nchunks, 1) * CSB_CHUNK;
    HIPCHK(s_val.alloc(sizeof(double) * ne));
    HIPCHK(s_idx.alloc(sizeof(unsigned) * ne));
    HIPCHK(s_cb.alloc(sizeof(int) * (size_t)std::max<long long>(nchunks, 1)));
    DevScratch s_gptr;
    HIPCHK(s_cptr.alloc(sizeof(long long) * cptr.size()));
    HIPCHK(s_gptr.alloc(sizeof(long long) * gptr.size()));
    probe_mem();
    HIPCHK(hipMemcpyAsync(s_cptr.p, cptr.data(), sizeof(long long) * cptr.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(s_gptr.p, gptr.data(), sizeof(long long) * gptr.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
    if (nchunks > 0)
        hipLaunchKernelGGL(k_csb_fill, dim3((unsigned)nchunks), dim3(CSB_CHUNK), 0, s,
                           (const unsigned *)s_perm.as<unsigned>(), rowk, colk, d_a, (const long long *)s_rbs.as<long long>(),
                           (const long long *)s_gptr.as<long long>(), (const int *)s_rst.as<int>(), nrb, NS, rmax,
                           (const short *)s_rexp.as<short>(), f32 ? 1 : 0, s_val.as<double>(), s_idx.as<unsigned>(),
                           s_cb.as<int>(), d_flags);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(got, d_flags, sizeof(got), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (got[3]) return LSQRHIP_OK;  // a chunk too wide for 18-bit local columns: not this layout
    if (got[2]) return LSQRHIP_OK;  // a value that does not survive its row's power of two exactly: not this layout
    // The narrow form of the index stream (csb.h: u16 rows + u8 column deltas, 11 bytes per nonzero), where every
    // delta fits a byte: LSQRHIP_CSB_NARROW=1.  Built and measured in round 4 (profiles/r04/csb_narrow_index.txt):
    // 8 % less memory, the SAME time per product on every BASELINE configuration (alternating runs in one process:
    // 3492 / 3499 against 3617 / 3475 us at config 4, 494 / 457 against 502 / 457 on one rank's block, 379 / 395
    // against 385 / 401 at config 5, 931 / 961 against 941 / 943 at config 3) -- what bounds the sweep is not the
    // bytes of its stream (scripts/csb_ceiling.hip).  So the 12-byte form stays the default: one code path fewer in
    // every product; the narrow one is for matrices that would not fit otherwise.
    bool narrow = false;
    DevScratch s_row16, s_del, s_cb4;
    if (nchunks > 0 && env_int("LSQRHIP_CSB_NARROW", 0) != 0) {
        HIPCHK(s_row16.alloc(sizeof(unsigned short) * ne));
        HIPCHK(s_del.alloc(ne));
        HIPCHK(s_cb4.alloc(sizeof(int) * (size_t)nchunks * CSB_U));
        HIPCHK(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(k_csb_narrow, dim3((unsigned)nchunks), dim3(CSB_CHUNK), 0, s, (const unsigned *)s_idx.as<unsigned>(),
                           (const int *)s_cb.as<int>(), s_row16.as<unsigned short>(), s_del.as<unsigned char>(),
                           s_cb4.as<int>(), d_flags);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(got, d_flags, sizeof(got), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        narrow = got[1] == 0;
        if (narrow) {
            s_idx.free_now();
            s_cb.free_now();
        } else {
            s_row16.free_now();
            s_del.free_now();
            s_cb4.free_now();
        }
    }
    out = Csr();
    out.rows = rows;
    out.cols = cols;
    out.rows_v = rows;
    out.csb = 1;
    out.cval = s_val.release<double>();
    out.cnarrow = narrow;
    {   // LSQRHIP_CSB_LOCKSTEP = 0 (the sweep of rounds 2-4: every wave on its own) | 1 | 2 (default) chunks per wave and step
        // Default by shape (profiles/r05/lockstep_ab.txt, lockstep_k_dense.txt: every shape in one process): ONE chunk per
        // wave and step, except long sweeps over sparse columns -- a unit (block x split) of >= 256 chunks per wave with
        // fewer than one nonzero per column and block (config 4 whole, the block of one rank of two: 2 chunks, -2 %).
        // Dense columns (config 3, 1000 per row) and short sweeps (a rank of eight's block, config 5) lose 1-10 % with 2.
        const int ls = env_int("LSQRHIP_CSB_LOCKSTEP", -1);
        if (ls >= 0) out.clockstep = ls > 2 ? 2 : ls;
        else {
            const double per_unit = (double)nnz / std::max(1.0, (double)nrb * std::max(S, 1));
            const double chunks_per_wave = per_unit / CSB_CHUNK / CSB_WAVES;
            const double per_column = (double)nnz / std::max(1, nrb) / std::max(cols, 1);
            out.clockstep = (chunks_per_wave >= 256.0 && per_column < 1.0) ? 2 : 1;
        }
        // the first form's second barrier, in front of the gathers: slower on every shape in the library (0.78 -> 0.84 ms at
        // config 3 per 100, 2.40 -> 2.62 at config 4: profiles/r05/lockstep_barrier_a.txt), kept as a knob for the A/B
        out.cbarrier_a = env_int("LSQRHIP_CSB_BARRIER_A", 0) != 0 ? 1 : 0;
        out.crounds = env_int("LSQRHIP_CSB_ROUNDS", 1) != 0 ? 1 : 0;
        const int sg = env_int("LSQRHIP_CSB_STAGGER", 0);
        out.cstagger = sg < 0 ? 0 : (sg > 64 ? 64 : sg);
    }
    if (narrow) {
        out.cidx = reinterpret_cast<unsigned *>(s_row16.release<unsigned short>());
        out.cdel = reinterpret_cast<unsigned *>(s_del.release<unsigned char>());
        out.ccb = s_cb4.release<int>();
    } else {
        out.cidx = s_idx.release<unsigned>();
        out.ccb = s_cb.release<int>();
    }
    out.cptr = s_cptr.release<long long>();
    out.crs = s_rst.release<int>();
    out.nchunks = nchunks;
    out.nstored = (int64_t)nchunks * CSB_CHUNK;
    out.nrb = nrb;
    out.R = rmax;   // the dummy accumulator's index (blocks hold at most this many rows)
    out.NS = NS;
    out.G = G;
    out.J = J;
    out.Pst = NS > 1 ? plan.P : 1;
    if (NS > 1) {
        out.gptr = s_gptr.release<long long>();
        out.phases = G;
    }
    if (!border.empty()) {
        HIPCHK(hipMalloc((void **)&out.border, sizeof(int) * border.size()));
        HIPCHK(hipMemcpy(out.border, border.data(), sizeof(int) * border.size(), hipMemcpyHostToDevice));
        out.phase_pos = phase_pos;
        out.phases = (int)phase_pos.size() - 1;
    }
    out.rexp = s_rexp.release<short>();
    HIPCHK(hipMalloc((void **)&out.zcoarse, sizeof(long long) * (size_t)rows));
    HIPCHK(hipMemsetAsync(out.zcoarse, 0, sizeof(long long) * (size_t)rows, s));
    out.S = S;
    if (S > 1) {
        HIPCHK(hipMalloc((void **)&out.zsplit, sizeof(long long) * (size_t)S * (size_t)rows));
        HIPCHK(hipMalloc((void **)&out.cbad, sizeof(int) * (size_t)nrb * CSB_QMAX));
        HIPCHK(hipMemsetAsync(out.cbad, 0, sizeof(int) * (size_t)nrb * CSB_QMAX, s));
        // the combine launch: enough workgroups for the whole chip (csb.h k_csb_combine)
        out.Q = (int)std::min<int64_t>(CSB_QMAX, std::max<int64_t>(1, (2 * CSB_GRID + nrb - 1) / nrb));
        while (out.Q > 1 && (int64_t)nrb * out.Q > SPMV_MAX_GRID) --out.Q;
        // (splits that exist only for the overlap plan: one workgroup per block, whose partial of sum y^2 is bit for
        // bit the unsplit kernel's -- a solve with the plan repeats the solve without it)
        if (S_plain == 1) out.Q = 1;
        // round 6: the split that arrives last at the block's ticket closes the block (csb.h) -- one partial per block,
        // the unsplit kernel's thread -> row mapping: norms bit for bit those of S = 1, and no second launch
        // ... where a block has TWO splits: the closer then pulls its own share and one other through its CU (config 5
        // transposed 320 -> 313 us, the block of a rank of four 642 -> 627).  From three splits on ONE closer's CU is the
        // bottleneck (0.85 MB through ~25 GB/s: 30 us for a full block at S = 4) and the combine launch, which spreads the same
        // bytes over the chip, stays 2-4 % ahead (a rank of eight's block 344 against 352-361 us:
        // profiles/r06/fuse_by_splits_and_harness_noise.txt).  LSQRHIP_CSB_FUSE=0 / 1 forces either.
        const int fenv = env_int("LSQRHIP_CSB_FUSE", -1);
        out.cfuse = fenv >= 0 ? (fenv != 0 ? 1 : 0) : (S == 2 ? 1 : 0);
        if (out.cfuse) out.Q = 1;
    }
    if (env_int("LSQRHIP_CSB_HAND", 1) != 0) {   // (=0: every launch of a product derives coefficients and grids itself, as in rounds 2-5)
        HIPCHK(hipMalloc(&out.chand, 64));
        HIPCHK(hipMemsetAsync(out.chand, 0, 64, s));
    }
    if (env_int("LSQRHIP_CSB_PROBE", 0) != 0) {
        HIPCHK(hipMalloc((void **)&out.cprobe, sizeof(unsigned long long) * CSB_PROBE_LAUNCHES * CSB_PROBE_WGS * 8));
        HIPCHK(hipMemsetAsync(out.cprobe, 0, sizeof(unsigned long long) * CSB_PROBE_LAUNCHES * CSB_PROBE_WGS * 8, s));
    }
    out.grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)nrb * S, CSB_GRID));  // workgroups per launch
    out.out_grid = nrb * out.Q;   // partials of sum(y^2) one product leaves behind
    out.nblk = nrb;
    // (+ 8 bytes per column: the k_csb_xmax pass reads the gathered vector once more than the sweeps' "x once";
    //  + 2 bytes per row: the rows' powers of two)
    out.bytes = (int64_t)nchunks * CSB_CHUNK * (narrow ? 11 : 12) + (int64_t)nchunks * (narrow ? 16 : 4) +
                (int64_t)(nrb + 1) * 8 + (int64_t)cols * 8 + (int64_t)rows * 2;
    return LSQRHIP_OK;
}

// sweep launches of one column-swept product (solve_loop.h launch_csb)
static int csb_sweep_launches(const Csr &c, bool rounds)
{
    const int S = std::max(c.S, 1);
    if (c.NS > 1) {   // stripes: every phase launches all blocks x its J splits
        const int step = std::max(1, c.grid / std::max(c.J, 1));
        return c.phases * std::max(1, (c.nrb + step - 1) / step);
    }
    const int step = rounds ? std::max(1, c.grid / S) : std::max(c.nrb, 1);
    if (c.border != nullptr) {   // segments: the rounds of every phase
        int nl = 0;
        for (int ph = 0; ph < c.phases; ++ph)
            nl += std::max(1, (c.phase_pos[(size_t)ph + 1] - c.phase_pos[(size_t)ph] + step - 1) / step);
        return nl;
    }
    return std::max(1, (c.nrb + step - 1) / step);
}

// Column panels for a product whose x vector has `cols` entries?  (spmv.h "Column panels")
//   LSQRHIP_PANELS    0 never | 1 whenever x exceeds one panel | unset: only if the columns are not local
//   LSQRHIP_PANEL_KB  panel size in KiB of x (default 2048: half an XCD's 4 MiB L2, the rest streams;
//                     swept 1024 / 2560 / 3584: profiles/r01/sweep_panels.txt)
//   LSQRHIP_XLDS      0 never | 1 whenever x exceeds one LDS panel | unset: scattered columns and
//                     >= 1.3 nonzeros per (row, LDS panel) -- panels of LSQRHIP_XLDS_COLS (7168) columns
//                     whose x slice lives in LDS (spmv.h XL): gathers stop being the bound
static void choose_panels(int rows, int cols, int64_t nnz, double mean_dev, int *panels, int *pw, int *xlds)
{
    *panels = 1;
    *pw = cols;
    *xlds = 0;
    const int mode = env_int("LSQRHIP_PANELS", -1);
    const int64_t kb = std::max(64, env_int("LSQRHIP_PANEL_KB", 2048));
    const int64_t width = (kb * 1024 / 8 + 1023) & ~(int64_t)1023;
    const bool scattered = mode == 1 || (mode != 0 && mean_dev >= 0.25 * (double)width);
    // LDS panels first: they win whenever the rows are dense enough to pay for many narrow panels
    const int xmode = env_int("LSQRHIP_XLDS", -1);
    if (xmode != 0) {
        const int64_t wl = std::min<int64_t>(XL_COLS, (std::max(1024, env_int("LSQRHIP_XLDS_COLS", XL_COLS)) + 1023) &
                                                           ~(int64_t)1023);
        const int64_t Pl = ((int64_t)cols + wl - 1) / wl;
        const bool fits = Pl > 1 && Pl * (int64_t)rows < (1ll << 31);
        // measured crossover against L2 panels (4M x 1M, r per row; ms per product L2 / LDS, both with
        // this round's kernels): r = 100 (0.7 per virtual row) 2.3 / 3.7, r = 150 (1.1) 3.4 / 3.7,
        // r = 200 (1.4) 4.6 / 4.3, r = 300 (2.1) 7.4 / 4.7, r = 600 14.8 / 8.0
        const bool dense = 10 * nnz >= 13 * Pl * (int64_t)rows;
        if (fits && (xmode == 1 || (scattered && dense && (int64_t)cols > 2 * width))) {
            *panels = (int)Pl;
            *pw = (int)wl;
            *xlds = 1;
            return;
        }
    }
    if (mode == 0) return;
    if ((int64_t)cols <= 2 * width) return;                  // x (nearly) fits L2 as it is
    if (!scattered) return;                                  // banded / local: plain CSR is better
    const int64_t P = ((int64_t)cols + width - 1) / width;
    if (P * (int64_t)rows >= (1ll << 31)) return;            // virtual rows must fit int32
    *panels = (int)P;
    *pw = (int)width;
}

// Column-swept row blocks (csb.h) are the layout of every matrix whose columns are scattered and which has
// enough nonzeros to keep 256 sweeping workgroups busy.  Measured against what the other rules would pick
// (profiles/r02/perf_csb_vs_xl.txt, perf_csb_small.txt; us per mode-1 product, other / CSB): 4M x 1M at 200 /
// 300 / 600 / 1000 per row (LDS panels) 3991 / 1744, 4575 / 2564, 6707 / 5041, 10109 / 8497; 1M x 1M x 500
// 1525 / 1040; 1M x 50k x 30 (row windows: x fits L2) 167 / 99, its transpose (LDS panels) 167 / 80; 2M x 500k
// x 10 145 / 83; 100k^2 x 100 53 / 45; 50k^2 x 100 29 / 26; 20k^2 x 200 24 / 20; 200k^2 x 50 53 / 56 (a tie);
// but 300k^2 x 8 = 2.4 M nonzeros 18 / 52: below ~4 M nonzeros the sweep is all fixed cost.
static bool csb_rule(int cols, int64_t nnz, double mean_dev)
{
    const int64_t kb = std::max(64, env_int("LSQRHIP_PANEL_KB", 2048));
    const double width = (double)((kb * 1024 / 8 + 1023) & ~(int64_t)1023);
    const bool scattered = mean_dev >= std::min(0.25 * width, (double)cols / 8.0);   // (uniform columns: cols / 3)
    return scattered && nnz >= (4ll << 20);
}

// work vectors, partial buffers, state: everything a solve needs besides the operator
static int alloc_workspace(H *h)
{
    hipStream_t s = h->stream;
    const size_t m1 = (size_t)std::max(h->m, 1), n1 = (size_t)std::max(h->n, 1);
    const size_t esz = h->f32 ? sizeof(float) : sizeof(double);
    HIPCHK(hipMalloc((void **)&h->U, esz * m1));
    HIPCHK(hipMalloc((void **)&h->V, esz * n1));
    HIPCHK(hipMalloc((void **)&h->W, esz * n1));
    HIPCHK(hipMalloc((void **)&h->X, esz * n1));
    HIPCHK(hipMalloc((void **)&h->SE, esz * n1));
    HIPCHK(hipMalloc((void **)&h->partials, sizeof(double) * 3 * SPMV_MAX_GRID));
    HIPCHK(hipMalloc((void **)&h->xmax_part, sizeof(double) * CSB_XMAX_GRID * (VEC_BLOCK / WAVE)));
    if (h->A.csb && h->AT.csb && env_int("LSQRHIP_CSB_XFOLD", 1) != 0) {   // solve_loop.h xmax_folded: two sets each
        const size_t bytes = sizeof(double) * 2 * CSB_XMAX_GRID * (VEC_BLOCK / WAVE);
        HIPCHK(hipMalloc((void **)&h->MXU, bytes));
        HIPCHK(hipMalloc((void **)&h->MXV, bytes));
        HIPCHK(hipMemsetAsync(h->MXU, 0, bytes, s));
        HIPCHK(hipMemsetAsync(h->MXV, 0, bytes, s));
    }
    {
        const int64_t zn = std::max<int64_t>(h->A.P > 1 ? h->A.rows_v : 0, h->AT.P > 1 ? h->AT.rows_v : 0);
        if (zn > 0) HIPCHK(hipMalloc((void **)&h->Z, sizeof(double) * (size_t)zn));
    }
    HIPCHK(hipMalloc((void **)&h->d_scalar, sizeof(double) * 4));
    for (double **pp : {&h->P1[0], &h->P1[1], &h->P2[0], &h->P2[1], &h->P3}) {
        HIPCHK(hipMalloc((void **)pp, sizeof(double) * SPMV_MAX_GRID));
        HIPCHK(hipMemsetAsync(*pp, 0, sizeof(double) * SPMV_MAX_GRID, s));
    }
    HIPCHK(hipMalloc((void **)&h->slots, sizeof(NormSlot) * 4));
    HIPCHK(hipMemsetAsync(h->slots, 0, sizeof(NormSlot) * 4, s));
    HIPCHK(hipMalloc((void **)&h->d_state, sizeof(LsqrState)));
    HIPCHK(hipHostMalloc((void **)&h->h_state, 3 * sizeof(LsqrState)));
    HIPCHK(hipHostMalloc((void **)&h->h_bslot, 64));
    for (hipEvent_t &e : h->ev_batch) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipMalloc((void **)&h->d_unit, sizeof(SpmvCoef)));
    HIPCHK(hipMalloc((void **)&h->d_zero, sizeof(int)));
    SpmvCoef unit{1.0, 1.0, 1.0, 0, 0};
    HIPCHK(hipMemcpyAsync(h->d_unit, &unit, sizeof(unit), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->d_zero, 0, sizeof(int), s));
    HIPCHK(hipEventCreate(&h->ev_loop0));
    HIPCHK(hipEventCreate(&h->ev_loop1));
    h->vgrid_m = vec_grid(h->m);
    h->vgrid_n = vec_grid(h->n);
    HIPCHK(hipStreamSynchronize(s));
    return LSQRHIP_OK;
}

// Value dictionary (valdict.h): h->dict / h->ndict when the matrix has <= 256 distinct values.
// LSQRHIP_VAL8=0 keeps 8-byte values.  `table` is scratch of >= VD_SLOTS words, `ctl` of 4 ints.
static int build_dictionary(H *h, const double *d_a, unsigned long long *table, int *ctl)
{
    h->ndict = 0;
    const int64_t nnz = h->nnz;
    if (nnz < VD_SLOTS || env_int("LSQRHIP_VAL8", 1) == 0) return LSQRHIP_OK;  // (scratch is nnz words)
    hipStream_t s = h->stream;
    HIPCHK(hipMemsetAsync(table, 0xFF, sizeof(unsigned long long) * VD_SLOTS, s));
    HIPCHK(hipMemsetAsync(ctl, 0, 4 * sizeof(int), s));
    const int g = (int)std::min<int64_t>((nnz + 255) / 256, 2048);
    hipLaunchKernelGGL(k_dict_collect, dim3(g), dim3(256), 0, s, d_a, nnz, table, ctl);
    HIPCHK(hipGetLastError());
    int got[4];
    HIPCHK(hipMemcpyAsync(got, ctl, sizeof(got), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (got[1] != 0 || got[0] < 1 || got[0] > VD_MAX) return LSQRHIP_OK;
    std::vector<unsigned long long> slots(VD_SLOTS), keys;
    HIPCHK(hipMemcpy(slots.data(), table, sizeof(unsigned long long) * VD_SLOTS, hipMemcpyDeviceToHost));
    for (unsigned long long v : slots)
        if (v != VD_EMPTY) keys.push_back(v);
    if ((int)keys.size() != got[0]) return LSQRHIP_OK;
    std::sort(keys.begin(), keys.end());
    keys.resize(VD_MAX, keys.back());  // padded: the SpMV loads all 256 entries
    HIPCHK(hipMalloc((void **)&h->dict, sizeof(double) * VD_MAX));
    HIPCHK(hipMemcpy(h->dict, keys.data(), sizeof(double) * VD_MAX, hipMemcpyHostToDevice));
    // every value must be in the table before a single code is written
    HIPCHK(hipMemsetAsync(ctl, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_dict_verify, dim3(g), dim3(256), 0, s, d_a, nnz, (const unsigned long long *)h->dict, got[0], ctl);
    HIPCHK(hipGetLastError());
    int missing = 1;
    HIPCHK(hipMemcpyAsync(&missing, ctl, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    h->ndict = missing ? 0 : got[0];
    return LSQRHIP_OK;
}

// REAL32 handle: the layout's 8-byte values become 4-byte ones (the build itself runs in binary64 on the
// exactly converted inputs; dictionary codes stay codes, the 256-entry table stays binary64).
static int values_to_f32(H *h, Csr &c)
{
    double **slot = c.csb ? &c.cval : (((c.sell == 1 && !c.sell_v8) || c.sell == 4) ? (double **)&c.sval : (!c.sell && c.val ? &c.val : nullptr));
    if (!slot || !*slot || c.nstored <= 0) return LSQRHIP_OK;
    float *f = nullptr;
    HIPCHK(hipMalloc((void **)&f, sizeof(float) * (size_t)c.nstored));
    const int g = (int)std::min<int64_t>((c.nstored + VEC_BLOCK - 1) / VEC_BLOCK, 65535);
    hipLaunchKernelGGL((k_convert<double, float>), dim3(g), dim3(VEC_BLOCK), 0, h->stream, (const double *)*slot, f,
                       c.nstored);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    (void)hipFree(*slot);
    *slot = reinterpret_cast<double *>(f);
    c.bytes -= 4 * c.nstored;
    return LSQRHIP_OK;
}

static int finish_create(H *h, const int *d_irow, const int *d_icol, const double *d_a)
{
    hipStream_t s = h->stream;
    const int64_t nnz = h->nnz;
    size_t free0 = 0, tot0 = 0;
    (void)hipMemGetInfo(&free0, &tot0);
    t_min_free = free0;
    DevScratch sA, sB, sH, sF;  // sort buffers, histogram (+ scan block sums), flags
    const int64_t nb = std::max<int64_t>((nnz + RS_TILE - 1) / RS_TILE, 1);
    HIPCHK(sA.alloc(sizeof(unsigned long long) * (size_t)std::max<int64_t>(nnz, 1)));
    HIPCHK(sB.alloc(sizeof(unsigned long long) * (size_t)std::max<int64_t>(nnz, 1)));
    HIPCHK(sH.alloc(sizeof(unsigned) * (256 * (size_t)nb + (256 * (size_t)nb) / SCAN_CHUNK + 64)));
    HIPCHK(sF.alloc(4 * sizeof(int)));
    unsigned long long *bufA = sA.as<unsigned long long>(), *bufB = sB.as<unsigned long long>();
    unsigned *hist = sH.as<unsigned>();
    int *d_flags = sF.as<int>();
    // locality of the column pattern (only looked at when a vector exceeds L2 or the matrix is large enough
    // for column-swept row blocks; indices that are out of range are caught by the build below, the measure
    // merely becomes meaningless)
    double mean_dev = 0.0;
    if (nnz > 0 && env_int("LSQRHIP_PANELS", -1) != 0 && (std::max(h->m, h->n) > 600000 || nnz >= (4ll << 20))) {
        unsigned long long *d_dev = (unsigned long long *)hist, dev = 0;
        HIPCHK(hipMemsetAsync(d_dev, 0, sizeof(dev), s));
        const int g = (int)std::min<int64_t>(std::max<int64_t>((nnz + 255) / 256, 1), 4096);
        hipLaunchKernelGGL(k_col_deviation, dim3(g), dim3(256), 0, s, d_irow, d_icol, nnz, h->m, h->n, d_dev);
        HIPCHK(hipMemcpyAsync(&dev, d_dev, sizeof(dev), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        mean_dev = (double)dev / (double)nnz;
    }
    RET(build_dictionary(h, d_a, bufA, d_flags));
    {   // 2^norm_exp > max|a_ij|: the power-of-two scale of the fused norms and of csb.h's binary grids
        double amax = 0.0;
        if (nnz > 0) {
            double *part = (double *)bufB;  // scratch of >= nnz words
            const int g = (int)std::min<int64_t>(std::max<int64_t>((nnz + 2 * VEC_BLOCK - 1) / (2 * VEC_BLOCK), 1),
                                                 std::min<int64_t>(VEC_MAX_GRID, nnz));
            std::vector<double> hp((size_t)g);
            hipLaunchKernelGGL(k_amax, dim3(g), dim3(VEC_BLOCK), 0, s, d_a, nnz, part);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(hp.data(), part, sizeof(double) * (size_t)g, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            for (double v : hp) amax = std::max(amax, v);
        }
        int e = 0;
        if (amax > 0.0 && amax < 1.0e308) (void)std::frexp(amax, &e);
        e = std::min(std::max(e, -1000), 1000);
        h->amax_exp = e;
        if (env_int("LSQRHIP_NORM_SCALE", 1) == 0) e = 0;  // ablation: plain sums of squares
        h->norm_exp = e;
        h->nsc.s = std::ldexp(1.0, -e);
        h->nsc.inv = std::ldexp(1.0, e);
    }
    int pa = 1, pwa = h->n, pt = 1, pwt = h->m, xa = 0, xt = 0;
    choose_panels(h->m, h->n, nnz, mean_dev, &pa, &pwa, &xa);                         // mode 1 gathers V (n)
    choose_panels(h->n, h->m, nnz, mean_dev * (double)std::max(h->m, 1) / (double)std::max(h->n, 1), &pt, &pwt, &xt);  // mode 2 gathers U (m)
    // Column-swept row blocks (csb.h) for scattered columns (csb_rule); a matrix the build declines (a block too
    // empty for 18-bit local columns) falls through to the panels / row windows chosen above.
    //   LSQRHIP_CSB   0 never | 1 for every matrix (tests) | 2 only instead of L2 panels (r02's first rule) | unset: csb_rule
    const int cmode = env_int("LSQRHIP_CSB", -1);
    const double dev_t = mean_dev * (double)std::max(h->m, 1) / (double)std::max(h->n, 1);
    // (a REAL32 handle has no panel kernels: scattered columns always go to the column-swept blocks)
    const bool csb_a = cmode == 1 || (cmode == 2 && pa > 1 && xa == 0) || (cmode < 0 && csb_rule(h->n, nnz, mean_dev)) ||
                       (h->f32 && pa > 1);
    const bool csb_t = cmode == 1 || (cmode == 2 && pt > 1 && xt == 0) || (cmode < 0 && csb_rule(h->m, nnz, dev_t)) ||
                       (h->f32 && pt > 1);
    // The overlap plan of the sharded engine (LSQRHIP_SHARD_OVERLAP=1; shard_engine.h): this matrix is one rank's row
    // block of a world of P -- the world lsqrhip_create_sharded is building, or LSQRHIP_SHARD_WORLD for a rank that
    // creates its own handle (bench.py --gpus N) -- whose n-vectors travel in G parts per slice (LSQRHIP_SHARD_PARTS).
    CsbPlan plan_a, plan_t;
    if (env_int("LSQRHIP_SHARD_OVERLAP", 0) != 0) {
        const int P = t_shard_world > 0 ? t_shard_world : env_int("LSQRHIP_SHARD_WORLD", 0);
        const int G = std::min(std::max(env_int("LSQRHIP_SHARD_PARTS", 2), 2), 4);
        if (P > 1 && (int64_t)P * G <= 256) {
            plan_a.P = plan_t.P = P;
            plan_a.G = plan_t.G = G;
            plan_a.stripes = true;    // mode 1 gathers v, which arrives part by part
            plan_t.segments = true;   // mode 2 produces T, which leaves part by part
        }
    }
    if (csb_a) RET(build_csb(s, d_irow, d_icol, d_a, nnz, h->m, h->n, h->f32, LSQRHIP_ERR_IROW, LSQRHIP_ERR_ICOL, sA, sB, hist, d_flags, h->A, plan_a));
    if (csb_t) RET(build_csb(s, d_icol, d_irow, d_a, nnz, h->n, h->m, h->f32, LSQRHIP_ERR_ICOL, LSQRHIP_ERR_IROW, sA, sB, hist, d_flags, h->AT, plan_t));
    if (!(h->A.csb && h->AT.csb)) {   // (build_csb releases the sort buffers while it fills: a row-window / panel build needs them again)
        const size_t sort_bytes = sizeof(unsigned long long) * (size_t)std::max<int64_t>(nnz, 1);
        if (!sA.p) HIPCHK(sA.alloc(sort_bytes));
        if (!sB.p) HIPCHK(sB.alloc(sort_bytes));
    }
    bufA = sA.as<unsigned long long>();
    bufB = sB.as<unsigned long long>();
    if (h->f32) {  // ... and row windows over the whole x where a block was too empty for them
        if (!h->A.csb) { pa = 1; pwa = h->n; xa = 0; }
        if (!h->AT.csb) { pt = 1; pwt = h->m; xt = 0; }
    }
    int rc = LSQRHIP_OK;
    if (h->off64) {
        if (!h->A.csb)
            rc = build_csr_T<long long>(s, d_irow, d_icol, d_a, nnz, h->m, h->n, LSQRHIP_ERR_IROW, LSQRHIP_ERR_ICOL, pa, pwa, xa, bufA, bufB, hist, d_flags, h->dict, h->ndict, h->A);
        if (rc == LSQRHIP_OK && !h->AT.csb)
            rc = build_csr_T<long long>(s, d_icol, d_irow, d_a, nnz, h->n, h->m, LSQRHIP_ERR_ICOL, LSQRHIP_ERR_IROW, pt, pwt, xt, bufA, bufB, hist, d_flags, h->dict, h->ndict, h->AT);
    } else {
        if (!h->A.csb)
            rc = build_csr_T<int>(s, d_irow, d_icol, d_a, nnz, h->m, h->n, LSQRHIP_ERR_IROW, LSQRHIP_ERR_ICOL, pa, pwa, xa, bufA, bufB, hist, d_flags, h->dict, h->ndict, h->A);
        if (rc == LSQRHIP_OK && !h->AT.csb)
            rc = build_csr_T<int>(s, d_icol, d_irow, d_a, nnz, h->n, h->m, LSQRHIP_ERR_ICOL, LSQRHIP_ERR_IROW, pt, pwt, xt, bufA, bufB, hist, d_flags, h->dict, h->ndict, h->AT);
    }
    dbg_stage(s, "both matrices built");
    probe_mem();
    RET(rc);
    if (h->f32) {
        RET(values_to_f32(h, h->A));
        RET(values_to_f32(h, h->AT));
    }
    {   // the matrix stream of the short-row layouts: non-temporal once an iteration's working set (both matrices and
        // the five vectors) exceeds TWICE the 256 MB Infinity Cache (common.h ld_stream); LSQRHIP_STREAM_NT=0 / 1 never /
        // always.  Measured crossover (profiles/r03/config2_patterns.txt section 15): at 288 MB plain loads still win
        // by 6 % (half the set survives an iteration), at 336 MB the two are equal, at 576 MB non-temporal wins by 11 %.
        const int64_t esz = h->f32 ? 4 : 8;
        const int64_t wset = h->A.bytes + h->AT.bytes + esz * ((int64_t)h->m + 4 * (int64_t)h->n);
        const int mode = env_int("LSQRHIP_STREAM_NT", -1);
        h->A.nt = h->AT.nt = mode < 0 ? wset > (512ll << 20) : mode != 0;
    }
    sA.free_now();
    sB.free_now();
    sH.free_now();
    RET(alloc_workspace(h));
    probe_mem();
    size_t free1 = 0;
    (void)hipMemGetInfo(&free1, &tot0);
    h->build_peak = (int64_t)free0 - (int64_t)t_min_free;
    h->build_kept = (int64_t)free0 - (int64_t)free1;
    return LSQRHIP_OK;
}

static int new_handle(int m, int n, int64_t nnz, H **out)
{
    if (!out) return fail(LSQRHIP_ERR_ARG, "null handle pointer");
    *out = nullptr;
    if (m < 0 || n < 0 || nnz < 0) return fail(LSQRHIP_ERR_ARG, "negative dimension");
    if (nnz >= (1ll << 32)) return fail(LSQRHIP_ERR_TOO_LARGE, lsqrhip_error_string(LSQRHIP_ERR_TOO_LARGE));
    RET(use_device());
    H *h = new H();
    h->loop_events = env_int("LSQRHIP_LOOP_EVENTS", h->loop_events);
    h->device = target_device();
    h->m = m;
    h->n = n;
    h->nnz = nnz;
    // 64-bit row pointers from 2^31 nonzeros on (LSQRHIP_OFF64=1 forces them: test hook for that path)
    h->off64 = nnz >= (1ll << 31) || env_int("LSQRHIP_OFF64", 0) != 0;
    // measurement hook: store values and vectors as float whatever entry point created the handle -- for
    // lsqrhip_bench_kernel on device-generated systems ONLY (solve entry points then move float vectors)
    h->f32 = env_int("LSQRHIP_STORE_F32", 0) != 0;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(LSQRHIP_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    h->stream = h->own_stream;
    *out = h;
    return LSQRHIP_OK;
}

static int tune_panel_grids(H *h);  // after solve_loop.h (needs the launchers)

extern "C" int lsqrhip_create_from_device_coo(int m, int n, int64_t nnz, const int *d_irow, const int *d_icol,
                                              const double *d_a, lsqrhip_handle_t *out)
{
    H *h = nullptr;
    RET(new_handle(m, n, nnz, &h));
    int rc = finish_create(h, d_irow, d_icol, d_a);
    if (rc == LSQRHIP_OK) rc = tune_panel_grids(h);
    if (rc != LSQRHIP_OK) {
        std::string keep = g_last_error;
        lsqrhip_destroy(h);
        g_last_error = keep;
        return rc;
    }
    *out = h;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_create(int m, int n, int64_t nnz, const int *irow, const int *icol, const double *a,
                              lsqrhip_handle_t *out)
{
    if (!out) return fail(LSQRHIP_ERR_ARG, "null handle pointer");
    *out = nullptr;
    if (nnz > 0 && (!irow || !icol || !a)) return fail(LSQRHIP_ERR_SIZES, lsqrhip_error_string(LSQRHIP_ERR_SIZES));
    RET(use_device());
    DevScratch sr, sc, sa;
    const size_t k = (size_t)std::max<int64_t>(nnz, 1);
    HIPCHK(sr.alloc(sizeof(int) * k));
    HIPCHK(sc.alloc(sizeof(int) * k));
    HIPCHK(sa.alloc(sizeof(double) * k));
    if (nnz > 0) {
        HIPCHK(hipMemcpy(sr.p, irow, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sc.p, icol, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sa.p, a, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
    }
    return lsqrhip_create_from_device_coo(m, n, nnz, sr.as<int>(), sc.as<int>(), sa.as<double>(), out);
}

extern "C" int lsqrhip_info(lsqrhip_handle_t h, int64_t *dims)
{
    if (!h || !dims) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (h->group) {  // sharded: the layouts of the first row block, the dimensions of the whole
        RET(lsqrhip_info(lsqrhip_group_rank0(h), dims));
        dims[0] = h->m;
        dims[1] = h->n;
        dims[2] = h->nnz;
        return LSQRHIP_OK;
    }
    dims[0] = h->m;
    dims[1] = h->n;
    dims[2] = h->nnz;
    dims[3] = h->A.bytes;
    dims[4] = h->AT.bytes;
    dims[5] = h->off64 ? 8 : 4;
    dims[6] = h->ndict;                          // value dictionary entries (0 = 8-byte values)
    dims[7] = h->A.sell == 3 ? 0 : ((h->A.val8 || h->A.sell_v8) ? 1 : 8);   // bytes per stored value (row patterns: none)
    dims[8] = (h->A.csb && h->A.cnarrow) ? 3 : ((h->A.col16 || h->A.sell_c16) ? 2 : 4);   // bytes per column index, CSR(A) (3: csb.h narrow form, u16 row + u8 column delta)
    dims[9] = (h->AT.csb && h->AT.cnarrow) ? 3 : ((h->AT.col16 || h->AT.sell_c16) ? 2 : 4);   //                        CSR(A')
    dims[10] = h->A.P;                           // column panels of CSR(A)
    dims[11] = h->AT.P;                          //                  CSR(A')
    dims[12] = h->A.sell;                        // sliced-ELL layout in use for A
    dims[13] = h->AT.sell;                       //                          for A'
    dims[14] = h->A.csb ? 3 : h->A.xlds;         // LDS-resident panels for A (3 = column-swept row blocks, csb.h)
    dims[15] = h->AT.csb ? 3 : h->AT.xlds;       //                     for A'
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// launch helpers, iteration schedules, solve_core
// ---------------------------------------------------------------------------
static int op_call(H *h, int mode, double *d_x, double *d_y);  // op_api.h
#include "solve_loop.h"

// Workgroups of a panelled product (row-window kernel over L2 column panels), chosen by timing.
// That kernel is bound by the vector-memory pipeline (PMC: texture addressers busy 76 %), and how
// many workgroups a CU should hold depends on the rows: uniform rows with 2-3 nonzeros per (row,
// panel) run up to 25 % FASTER with 4 workgroups per CU than with 8 (config 4: 10.4 -> 7.8 ms per
// product), skewed rows (power law, mode 1) up to 50 % slower.  So each panelled matrix is tried
// at a few grid sizes once (1 + 2 launches each) and keeps the fastest.  A panelled product's
// results do not depend on the grid (per-panel row sums, then k_panel_combine with its own grid),
// so the choice can never change a bit.  LSQRHIP_SPMV_GRID fixes the grid, LSQRHIP_TUNE=0 skips.
static int tune_one_panel_grid(H *h, Csr &c, const double *x, double *y)
{
    if (c.P <= 1 || c.sell || c.xlds == 2 || c.nblk <= 0) return LSQRHIP_OK;
    if (env_int("LSQRHIP_SPMV_GRID", 0) > 0 || env_int("LSQRHIP_TUNE", 1) == 0) return LSQRHIP_OK;
    hipStream_t s = h->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    const int original = c.grid;
    int best = original;
    float best_ms = 0.f;
    int rc = LSQRHIP_OK;
    for (int cand : {SPMV_MAX_GRID, 1536, 1024, 768}) {
        int64_t g = std::min<int64_t>(c.nblk, cand);
        if (g >= 8) g &= ~(int64_t)7;
        if (g < 1 || (cand != SPMV_MAX_GRID && g == original)) continue;
        c.grid = (int)g;
        SpmvArgs a;
        a.c = &c; a.x = x; a.y = y; a.coef = h->d_unit; a.stop = h->d_zero; a.pout = h->partials; a.stream = s;
        launch_spmv_args(h, a);  // warm
        if (hipEventRecord(e0, s) != hipSuccess) { rc = LSQRHIP_ERR_HIP; break; }
        launch_spmv_args(h, a);
        launch_spmv_args(h, a);
        if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipGetLastError() != hipSuccess) { rc = LSQRHIP_ERR_HIP; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (best_ms == 0.f || ms < 0.97f * best_ms) {  // a smaller grid must win by 3 %
            best_ms = ms;
            best = (int)g;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c.grid = rc == LSQRHIP_OK ? best : original;  // best effort: a failed measurement keeps the default
    return LSQRHIP_OK;
}

static int tune_panel_grids(H *h)
{
    if (h->A.P <= 1 && h->AT.P <= 1) return LSQRHIP_OK;
    hipStream_t s = h->stream;
    HIPCHK(hipMemsetAsync(h->U, 0, sizeof(double) * (size_t)std::max(h->m, 1), s));
    HIPCHK(hipMemsetAsync(h->V, 0, sizeof(double) * (size_t)std::max(h->n, 1), s));
    RET(tune_one_panel_grid(h, h->A, h->V, h->U));   // mode 1: y (m) += A x (n)
    RET(tune_one_panel_grid(h, h->AT, h->U, h->V));  // mode 2: x (n) += A' y (m)
    HIPCHK(hipStreamSynchronize(s));
    return LSQRHIP_OK;
}

#define NOT_F32(h)                                                                                               \
    do {                                                                                                         \
        if ((h) && (h)->f32)                                                                                     \
            return fail(LSQRHIP_ERR_ARG, "REAL32 handle: use lsqrhip_solve_f32 / lsqrhip_aprod_f32 (float vectors)"); \
    } while (0)

extern "C" int lsqrhip_solve(lsqrhip_handle_t h, const double *b, double damp, double atol, double btol,
                             double conlim, int itnlim, int wantse, int want_log, double *x, double *se, int *istop,
                             int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    NOT_F32(h);
    return solve_core(h, b, false, damp, atol, btol, conlim, itnlim, wantse, want_log, x, se, false, istop, itn,
                      anorm, acond, rnorm, arnorm, xnorm);
}

extern "C" int lsqrhip_solve_device(lsqrhip_handle_t h, const double *d_b, double damp, double atol, double btol,
                                    double conlim, int itnlim, int wantse, int want_log, double *d_x, double *d_se,
                                    int *istop, int *itn, double *anorm, double *acond, double *rnorm,
                                    double *arnorm, double *xnorm)
{
    NOT_F32(h);
    return solve_core(h, d_b, true, damp, atol, btol, conlim, itnlim, wantse, want_log, d_x, d_se, true, istop, itn,
                      anorm, acond, rnorm, arnorm, xnorm);
}

// ---------------------------------------------------------------------------
// aprod
// ---------------------------------------------------------------------------
extern "C" int lsqrhip_aprod_device(lsqrhip_handle_t h, int mode, double *d_x, double *d_y)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (h->f32 && !h->f32_device_ok) NOT_F32(h);
    if (mode != 1 && mode != 2) return fail(LSQRHIP_ERR_MODE, lsqrhip_error_string(LSQRHIP_ERR_MODE));
    if (h->group) return fail(LSQRHIP_ERR_ARG, "a sharded handle takes host vectors: lsqrhip_aprod");
    HIPCHK(hipSetDevice(h->device));
    if (h->op) {  // user device operator (op_api.h)
        RET(op_call(h, mode, d_x, d_y));
        HIPCHK(hipStreamSynchronize(h->stream));
        return LSQRHIP_OK;
    }
    if (mode == 1) launch_spmv(h, h->A, d_x, d_y, h->d_unit, h->d_zero);   // y += A x
    else launch_spmv(h, h->AT, d_y, d_x, h->d_unit, h->d_zero);            // x += A' y
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_aprod(lsqrhip_handle_t h, int mode, double *x, double *y)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    NOT_F32(h);
    if (mode != 1 && mode != 2) return fail(LSQRHIP_ERR_MODE, lsqrhip_error_string(LSQRHIP_ERR_MODE));
    if (h->group) return aprod_group_host(h, mode, x, y);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // borrow the solver's work vectors: V (n) for x, U (m) for y
    if (h->n > 0) HIPCHK(hipMemcpyAsync(h->V, x, sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, s));
    if (h->m > 0) HIPCHK(hipMemcpyAsync(h->U, y, sizeof(double) * (size_t)h->m, hipMemcpyHostToDevice, s));
    RET(lsqrhip_aprod_device(h, mode, h->V, h->U));
    if (mode == 1 && h->m > 0) HIPCHK(hipMemcpyAsync(y, h->U, sizeof(double) * (size_t)h->m, hipMemcpyDeviceToHost, s));
    if (mode == 2 && h->n > 0) HIPCHK(hipMemcpyAsync(x, h->V, sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// device BLAS-1 (unit stride)
// ---------------------------------------------------------------------------
static int dev_dot(H *h, int64_t n, const double *d_x, const double *d_y, double *result)
{
    *result = 0.0;
    if (n <= 0) return LSQRHIP_OK;
    const int g = vec_grid(n);
    hipLaunchKernelGGL(k_dot, dim3(g), dim3(VEC_BLOCK), 0, h->stream, d_x, d_y, n, h->partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(VEC_BLOCK), 0, h->stream, (const double *)h->partials, g,
                       h->d_scalar);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(result, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_dnrm2(lsqrhip_handle_t h, int64_t n, const double *d_x, double *result)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h || !result) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    HIPCHK(hipSetDevice(h->device));
    // Stand-alone dnrm2 (src/lsqrblas.f90:123-159 is a scaled sum of squares): two passes, the
    // largest magnitude and then the sum of (x * 2^-e)^2 with 2^e ~ max|x|, so that neither
    // x^2 overflow nor underflow is possible -- same range as the reference's dlassq form.
    // (Inside the iteration the norms are fused, unscaled sums: u, v are renormalised every step.)
    *result = 0.0;
    if (n < 1) return LSQRHIP_OK;
    const int g = vec_grid(2 * n);
    std::vector<double> part((size_t)g);
    hipLaunchKernelGGL(k_amax, dim3(g), dim3(VEC_BLOCK), 0, h->stream, d_x, n, h->partials);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(part.data(), h->partials, sizeof(double) * (size_t)g, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double amax = 0.0;
    for (double v : part) amax = std::max(amax, v);      // NaN-free inputs assumed, like the reference
    if (!(amax > 0.0)) return LSQRHIP_OK;
    if (std::isinf(amax)) {
        *result = amax;
        return LSQRHIP_OK;
    }
    int e = 0;
    (void)std::frexp(amax, &e);                           // amax = f * 2^e, f in [0.5, 1)
    const double sc = std::ldexp(1.0, -e);
    hipLaunchKernelGGL(k_sumsq_scaled, dim3(g), dim3(VEC_BLOCK), 0, h->stream, d_x, n, sc, h->partials);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(VEC_BLOCK), 0, h->stream, (const double *)h->partials, g,
                       h->d_scalar);
    HIPCHK(hipGetLastError());
    double ss = 0.0;
    HIPCHK(hipMemcpyAsync(&ss, h->d_scalar, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *result = std::ldexp(std::sqrt(ss), e);
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_ddot(lsqrhip_handle_t h, int64_t n, const double *d_x, const double *d_y, double *result)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h || !result) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    HIPCHK(hipSetDevice(h->device));
    return dev_dot(h, n, d_x, d_y, result);
}

extern "C" int lsqrhip_dscal(lsqrhip_handle_t h, int64_t n, double da, double *d_x)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (n <= 0) return LSQRHIP_OK;
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_scale, dim3(vec_grid(n)), dim3(VEC_BLOCK), 0, h->stream, d_x, n, da);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_dcopy(lsqrhip_handle_t h, int64_t n, const double *d_x, double *d_y)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (n <= 0) return LSQRHIP_OK;
    HIPCHK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_copy, dim3(vec_grid(n)), dim3(VEC_BLOCK), 0, h->stream, d_x, d_y, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// acheck / xcheck on the device operator
// ---------------------------------------------------------------------------
struct DevVec {
    double *p = nullptr;
    ~DevVec()
    {
        if (p) (void)hipFree(p);
    }
    int alloc(int64_t n)
    {
        hipError_t e = hipMalloc((void **)&p, sizeof(double) * (size_t)std::max<int64_t>(n, 1));
        return e == hipSuccess ? LSQRHIP_OK : fail(LSQRHIP_ERR_ALLOC, hipGetErrorString(e));
    }
};

extern "C" int lsqrhip_acheck(lsqrhip_handle_t h, double eps, int *inform, double *relerr)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h || !inform) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    NOT_F32(h);
    HIPCHK(hipSetDevice(h->device));
    const int64_t m = h->m, n = h->n;
    hipStream_t s = h->stream;
    DevVec v, w, x, y;
    RET(v.alloc(n)); RET(w.alloc(m)); RET(x.alloc(n)); RET(y.alloc(m));
    const double tol = std::pow(eps, 0.5);                                      // src/lsqr.f90:939
    hipLaunchKernelGGL(k_acheck_fill, dim3(vec_grid(n)), dim3(VEC_BLOCK), 0, s, x.p, n, 0);   // :946-950
    hipLaunchKernelGGL(k_acheck_fill, dim3(vec_grid(m)), dim3(VEC_BLOCK), 0, s, y.p, m, 1);   // :952-956
    double alfa = 0, beta = 0;
    RET(lsqrhip_dnrm2(h, n, x.p, &alfa));                                       // :958-961
    RET(lsqrhip_dnrm2(h, m, y.p, &beta));
    RET(lsqrhip_dscal(h, n, 1.0 / alfa, x.p));
    RET(lsqrhip_dscal(h, m, 1.0 / beta, y.p));
    RET(lsqrhip_dcopy(h, m, y.p, w.p));                                         // :969-972
    RET(lsqrhip_dcopy(h, n, x.p, v.p));
    RET(lsqrhip_aprod_device(h, 1, x.p, w.p));
    RET(lsqrhip_aprod_device(h, 2, v.p, y.p));
    RET(lsqrhip_ddot(h, m, y.p, w.p, &alfa));                                   // :976-980
    RET(lsqrhip_ddot(h, n, x.p, v.p, &beta));
    const double test1 = std::fabs(alfa - beta);
    const double test2 = 1.0 + std::fabs(alfa) + std::fabs(beta);
    const double test3 = test1 / test2;
    if (relerr) *relerr = test3;
    *inform = test3 <= tol ? 0 : 1;                                             // :984-992
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_xcheck(lsqrhip_handle_t h, double anorm, double damp, double eps, const double *b,
                              const double *x, double *u, double *v, double *w, int *inform, double *tests)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h || !inform || !tests) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    NOT_F32(h);
    HIPCHK(hipSetDevice(h->device));
    const int64_t m = h->m, n = h->n;
    hipStream_t s = h->stream;
    DevVec db, dx, du, dv, dw;
    RET(db.alloc(m)); RET(dx.alloc(n)); RET(du.alloc(m)); RET(dv.alloc(n)); RET(dw.alloc(n));
    if (m > 0) HIPCHK(hipMemcpyAsync(db.p, b, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, s));
    if (n > 0) HIPCHK(hipMemcpyAsync(dx.p, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    const double dampsq = damp * damp, tol = std::pow(eps, 0.5);
    RET(lsqrhip_dcopy(h, m, db.p, du.p));                                       // :1073-1076
    RET(lsqrhip_dscal(h, m, -1.0, du.p));
    RET(lsqrhip_aprod_device(h, 1, dx.p, du.p));
    RET(lsqrhip_dscal(h, m, -1.0, du.p));
    if (n > 0) HIPCHK(hipMemsetAsync(dv.p, 0, sizeof(double) * (size_t)n, s));  // :1080-1083
    RET(lsqrhip_aprod_device(h, 2, dv.p, du.p));
    RET(lsqrhip_dcopy(h, n, dv.p, dw.p));                                       // :1089-1094
    if (damp != 0.0 && n > 0) {
        hipLaunchKernelGGL(k_axpy, dim3(vec_grid(n)), dim3(VEC_BLOCK), 0, s, dw.p, (const double *)dx.p, n, -dampsq);
        HIPCHK(hipGetLastError());
    }
    double bnorm, xnorm, rho1, sigma1, rho2, sigma2;
    RET(lsqrhip_dnrm2(h, m, db.p, &bnorm));                                     // :1098-1101
    RET(lsqrhip_dnrm2(h, n, dx.p, &xnorm));
    RET(lsqrhip_dnrm2(h, m, du.p, &rho1));
    RET(lsqrhip_dnrm2(h, n, dv.p, &sigma1));
    if (damp == 0.0) {                                                          // :1110-1124
        rho2 = rho1;
        sigma2 = sigma1;
    } else {
        rho2 = std::sqrt(rho1 * rho1 + dampsq * (xnorm * xnorm));
        RET(lsqrhip_dnrm2(h, n, dw.p, &sigma2));
    }
    double test1, test2, test3;
    if (bnorm == 0.0 && xnorm == 0.0) {                                         // :1129-1144
        *inform = 0;
        test1 = test2 = test3 = 0.0;
    } else {
        *inform = 4;
        test1 = rho1 / (bnorm + anorm * xnorm);
        test2 = 0.0;
        if (rho1 > 0.0) test2 = sigma1 / (anorm * rho1);
        test3 = test2;
        if (rho2 > 0.0) test3 = sigma2 / (anorm * rho2);
        if (test3 <= tol) *inform = 3;
        if (test2 <= tol) *inform = 2;
        if (test1 <= tol) *inform = 1;
    }
    tests[0] = test1;
    tests[1] = test2;
    tests[2] = test3;
    if (u && m > 0) HIPCHK(hipMemcpyAsync(u, du.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
    if (v && n > 0) HIPCHK(hipMemcpyAsync(v, dv.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    if (w && n > 0) HIPCHK(hipMemcpyAsync(w, dw.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// kernel micro-timing (bench.py roofline leg)
// ---------------------------------------------------------------------------
extern "C" int lsqrhip_bench_kernel(lsqrhip_handle_t h, int which, int reps, double *avg_ms)
{
    if (h && h->group) return fail(LSQRHIP_ERR_ARG, "not available on a handle sharded over several GPUs (lsqrhip_create_sharded)");
    if (!h || !avg_ms || reps < 1 || which < 1 || which > 3) return fail(LSQRHIP_ERR_ARG, "bad bench_kernel arguments");
    if (h->op && which != 3) return fail(LSQRHIP_ERR_ARG, "operator handles have no SpMV kernel to time");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // operands: finite, small; coefficients that keep them bounded over `reps` launches
    if (h->f32) {  // (float vectors: zeros will do, no kernel here is data dependent)
        HIPCHK(hipMemsetAsync(h->U, 0, sizeof(float) * (size_t)std::max(h->m, 1), s));
        for (double *p : {h->V, h->W, h->X}) HIPCHK(hipMemsetAsync(p, 0, sizeof(float) * (size_t)std::max(h->n, 1), s));
    } else {
        hipLaunchKernelGGL(k_fill, dim3(h->vgrid_m), dim3(VEC_BLOCK), 0, s, h->U, (int64_t)h->m, 1.0e-3);
        hipLaunchKernelGGL(k_fill, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->V, (int64_t)h->n, 1.0e-3);
        hipLaunchKernelGGL(k_fill, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->W, (int64_t)h->n, 1.0e-3);
        hipLaunchKernelGGL(k_fill, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->X, (int64_t)h->n, 0.0);
    }
    LsqrState tmp;
    std::memset(&tmp, 0, sizeof(tmp));
    tmp.t1 = 1.0e-3; tmp.t2 = -0.5; tmp.t3 = 1.0e-3; tmp.sv = 1.0; tmp.su = 1.0;
    tmp.c1.sx = 1.0; tmp.c1.sy = 0.5; tmp.c1.cy = -0.5;   // y <- -0.25 y + A x : bounded
    tmp.c2 = tmp.c1;
    DevScratch s_tmp;  // released on every exit path
    HIPCHK(s_tmp.alloc(sizeof(LsqrState)));
    LsqrState *d_tmp = s_tmp.as<LsqrState>();
    HIPCHK(hipMemcpyAsync(d_tmp, &tmp, sizeof(tmp), hipMemcpyHostToDevice, s));
    // A column-swept product inside the solver's loop runs no max|x| pass: the product that wrote x left the piece
    // maxima, and this one leaves those of y (solve_loop.h xmax_folded).  Timed here in that form: the operand is a
    // constant vector, its piece maxima are written down directly (the word csb_hi_up would keep), the epilogue's
    // share stays in.
    const bool loop_form = which != 3 && xmax_folded(h);
    if (loop_form) {
        double bound = 0.0;
        if (!h->f32) {
            const double v = 1.0e-3;
            unsigned long long bits;
            std::memcpy(&bits, &v, sizeof bits);
            bits = ((bits >> 32) + 1ull) << 32;
            std::memcpy(&bound, &bits, sizeof bound);
        }
        hipLaunchKernelGGL(k_fill, dim3(vec_grid(CSB_XMAX_PIECES)), dim3(VEC_BLOCK), 0, s, h->xmax_part,
                           (int64_t)CSB_XMAX_PIECES, bound);
    }
    auto one = [&]() {
        if (which == 1 || which == 2) {
            SpmvArgs a;
            a.c = which == 1 ? &h->A : &h->AT;
            a.x = which == 1 ? h->V : h->U;
            a.y = which == 1 ? h->U : h->V;
            a.coef = which == 1 ? &d_tmp->c1 : &d_tmp->c2;
            a.stop = &d_tmp->stop; a.pout = h->partials; a.stream = s; a.unit_x = true;
            if (loop_form) {
                a.xmax_in = h->xmax_part;
                a.nxmax_in = csb_npieces(a.c->cols);
                a.ymax_out = which == 1 ? h->MXU : h->MXV;   // (never cleared here: a solve zeroes the sets when it starts)
            }
            launch_spmv_args(h, a);
        }
        else if (h->f32)
            hipLaunchKernelGGL(k_update<float>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, (float *)h->X, (float *)h->W,
                               (const float *)h->V, (float *)h->SE, (int64_t)h->n, (const LsqrState *)d_tmp, h->partials);
        else
            hipLaunchKernelGGL(k_update<double>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->X, h->W,
                               (const double *)h->V, h->SE, (int64_t)h->n, (const LsqrState *)d_tmp, h->partials);
    };
    for (int i = 0; i < 3; ++i) one();  // warm
    HIPCHK(hipEventRecord(h->ev_loop0, s));
    for (int i = 0; i < reps; ++i) one();
    HIPCHK(hipEventRecord(h->ev_loop1, s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_loop0, h->ev_loop1));
    *avg_ms = (double)ms / reps;
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// log, timing, options, memory helpers
// ---------------------------------------------------------------------------
// (a handle sharded over several GPUs by this process keeps its log on rank 0's sub-handle: shard_engine.h)
static H *log_owner(H *h);

extern "C" int lsqrhip_log_count(lsqrhip_handle_t h) { return h ? log_owner(h)->log_count : 0; }

extern "C" int lsqrhip_log_fetch(lsqrhip_handle_t h, int first, int count, double *records)
{
    if (!h || !records) return fail(LSQRHIP_ERR_ARG, "null handle or buffer");
    h = log_owner(h);
    if (first < 0 || count < 0 || first + count > h->log_count) return fail(LSQRHIP_ERR_ARG, "log range out of bounds");
    std::memcpy(records, h->h_log.data() + (size_t)first * LOG_STRIDE, sizeof(double) * LOG_STRIDE * (size_t)count);
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_log_extras(lsqrhip_handle_t h, double *out)
{
    if (h) h = log_owner(h);
    if (!h || !out || !h->h_state) return fail(LSQRHIP_ERR_ARG, "null handle or buffer");
    const LsqrState &r = *h->h_state;
    out[0] = r.bnorm;
    out[1] = r.dxmax;
    out[2] = (double)r.maxdx;
    out[3] = r.alpha0;
    out[4] = r.beta0;
    out[5] = r.test2_0;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_last_timing(lsqrhip_handle_t h, lsqrhip_timing_t *t)
{
    if (!h || !t) return fail(LSQRHIP_ERR_ARG, "null handle or buffer");
    *t = h->timing;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_set_option(lsqrhip_handle_t h, const char *name, int64_t value)
{
    if (!h || !name) return fail(LSQRHIP_ERR_ARG, "null handle or option name");
    const std::string k(name);
    if (k == "graph") h->use_graph = value != 0;
    else if (k == "graph_iters") {
        if (value < 1 || value > 1024) return fail(LSQRHIP_ERR_ARG, "graph_iters must be in [1,1024]");
        h->graph_iters = (int)value;
    } else if (k == "time_kernels") h->time_kernels = value != 0;
    else if (k == "poll_ahead") h->poll_ahead = value != 0;
    else if (k == "loop_events") h->loop_events = value != 0;
    else if (k == "op_batch") h->op_batch = value < 1 ? 1 : (int)value;
    else if (k == "pipeline") h->pipeline = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    else if (k == "shard_log") h->shard.want_log = value != 0;   // (rank 0 of a one-process-per-GPU world)
    else if (k == "norm_exp") {  // the ranks of a row-sharded solve must scale their sums of squares alike
        if (value < -1000 || value > 1000) return fail(LSQRHIP_ERR_ARG, "norm_exp must be in [-1000, 1000]");
        // captured kernel nodes hold the scale BY VALUE (SpmvArgs.nsc, k_update_lazy): batches captured under
        // another exponent would sum (y * old scale)^2 and rescale by the new one -- they are rebuilt
        if ((int)value != h->norm_exp) {
            h->graph_dirty = true;
            ++h->graph_epoch;
        }
        h->norm_exp = (int)value;
        h->nsc.s = std::ldexp(1.0, -(int)value);
        h->nsc.inv = std::ldexp(1.0, (int)value);
    } else return fail(LSQRHIP_ERR_ARG, "unknown option: " + k);
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_get_option(lsqrhip_handle_t h, const char *name, int64_t *value)
{
    if (!h || !name || !value) return fail(LSQRHIP_ERR_ARG, "null handle, option name or result");
    const std::string k(name);
    if (k == "graph") *value = h->use_graph;
    else if (k == "graph_iters") *value = h->graph_iters;
    else if (k == "time_kernels") *value = h->time_kernels;
    else if (k == "poll_ahead") *value = h->poll_ahead;
    else if (k == "loop_events") *value = h->loop_events;
    else if (k == "op_batch") *value = h->op_batch;
    else if (k == "pipeline") *value = h->pipeline;
    else if (k == "norm_exp") *value = h->norm_exp;
    else if (k == "build_peak_bytes") *value = h->build_peak;   // peak device memory of the create beyond the caller's triplets
    else if (k == "build_kept_bytes") *value = h->build_kept;   // ... and what the handle holds now (layouts + work vectors)
    else if (k == "log_truncated") *value = log_owner(h)->h_state ? log_owner(h)->h_state->log_truncated : 0;  // of the last solve
    else if (k == "launches_mode1" || k == "launches_mode2" || k == "dispatches_mode1" || k == "dispatches_mode2") {
        // launches_*: launches of the product's main kernel (+ its combine kernel) -- what rocprofv3's per-kernel
        // average is divided over; dispatches_*: every kernel one product enqueues (the max|x| pass of csb.h too)
        const Csr &c = (k == "launches_mode1" || k == "dispatches_mode1") ? h->A : h->AT;
        const bool all = k[0] == 'd';
        const int rounds = c.crounds;
        // (the pass is not part of a product inside the loop when both matrices are column-swept: xmax_folded, and
        //  lsqrhip_bench_kernel times that form)
        if (c.csb) *value = (all && !xmax_folded(h) ? 1 : 0) + csb_sweep_launches(c, rounds != 0) + (c.S > 1 && !c.cfuse ? 1 : 0);
        else *value = c.P > 1 ? 2 : 1;
    } else if (k == "csb_blocks_mode1" || k == "csb_blocks_mode2") {  // row blocks of a column-swept layout (0: another layout)
        const Csr &c = k == "csb_blocks_mode1" ? h->A : h->AT;
        *value = c.csb ? c.nrb : 0;
    } else if (k == "csb_phases_mode1" || k == "csb_phases_mode2") {   // launch phases of the overlap plan (1: none)
        const Csr &c = k == "csb_phases_mode1" ? h->A : h->AT;
        *value = c.csb ? std::max(c.phases, 1) : 0;
    } else if (k == "csb_splits_mode1" || k == "csb_splits_mode2") {
        const Csr &c = k == "csb_splits_mode1" ? h->A : h->AT;
        *value = c.csb ? c.S : 0;
    } else if (k == "csb_probe_mode1" || k == "csb_probe_mode2") {   // device address of the phase clocks (0: LSQRHIP_CSB_PROBE was not set)
        const Csr &c = k == "csb_probe_mode1" ? h->A : h->AT;
        *value = (int64_t)(uintptr_t)c.cprobe;
    } else if (k == "csb_fuse_mode1" || k == "csb_fuse_mode2") {   // column splits closed by their last arriver (no combine launch)
        const Csr &c = k == "csb_fuse_mode1" ? h->A : h->AT;
        *value = c.csb && c.S > 1 ? c.cfuse : 0;
    } else if (k == "pat_pair_mode1" || k == "pat_pair_mode2") {   // row patterns in the paired-rows form (pat.h)
        const Csr &c = k == "pat_pair_mode1" ? h->A : h->AT;
        *value = c.pat_pair ? 1 : 0;
    } else if (k == "pat_wide_mode1" || k == "pat_wide_mode2") {   // patterns of the wide row-pattern table in use (0: another layout)
        const Csr &c = k == "pat_wide_mode1" ? h->A : h->AT;
        *value = c.pat_wide ? c.npat : 0;
    } else if (k == "csb_lockstep_mode1" || k == "csb_lockstep_mode2") {   // chunks per wave and lock-step step (0: free-running)
        const Csr &c = k == "csb_lockstep_mode1" ? h->A : h->AT;
        *value = c.csb ? c.clockstep : 0;
    } else if (k == "shard_engine_flags") {
        // what the C++ engine left in this rank's stage context (0 between solves -- also after an engine solve that failed):
        // 1 own slice read in T | 2 `sums` is the long message | 4 norms gathered by the engine | 8 a sharded solve is open
        const ShardCtx &c = h->shard;
        *value = (c.own_in_T ? 1 : 0) | (c.vmax_msg ? 2 : 0) | (c.gath != nullptr ? 4 : 0) | (c.active ? 8 : 0);
    } else if (k == "shard_overlap" || k == "shard_parts" || k == "shard_copy") {
        // the schedule the engine of this handle REALLY runs (a requested overlap that could not be set up is off here;
        // shard_copy: the n-vector exchanges are copies between mapped buffers, not RCCL send / receive kernels)
        *value = shard_effective(h, k == "shard_overlap" ? 0 : (k == "shard_parts" ? 1 : 2));
    } else return fail(LSQRHIP_ERR_ARG, "unknown option: " + k);
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_set_stream(lsqrhip_handle_t h, void *hip_stream)
{
    if (!h) return fail(LSQRHIP_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    destroy_graph(h);
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_dev_alloc(void **d_ptr, int64_t bytes)
{
    if (!d_ptr || bytes < 0) return fail(LSQRHIP_ERR_ARG, "bad alloc request");
    RET(use_device());
    HIPCHK(hipMalloc(d_ptr, (size_t)std::max<int64_t>(bytes, 8)));
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_dev_free(void *d_ptr)
{
    if (d_ptr) HIPCHK(hipFree(d_ptr));
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_dev_upload(void *d_dst, const void *src, int64_t bytes)
{
    if (bytes > 0) HIPCHK(hipMemcpy(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice));
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_dev_download(void *dst, const void *d_src, int64_t bytes)
{
    if (bytes > 0) HIPCHK(hipMemcpy(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_dev_sync(void)
{
    HIPCHK(hipDeviceSynchronize());
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// REAL32: the reference's precision macro (src/lsqr_kinds.F90:16-17: wp = real32)
// ---------------------------------------------------------------------------
extern "C" int lsqrhip_create_f32(int m, int n, int64_t nnz, const int *irow, const int *icol, const float *a,
                                  lsqrhip_handle_t *out)
{
    if (!out) return fail(LSQRHIP_ERR_ARG, "null handle pointer");
    *out = nullptr;
    if (nnz > 0 && (!irow || !icol || !a)) return fail(LSQRHIP_ERR_SIZES, lsqrhip_error_string(LSQRHIP_ERR_SIZES));
    RET(use_device());
    DevScratch sr, sc, sa, sf;
    const size_t k = (size_t)std::max<int64_t>(nnz, 1);
    HIPCHK(sr.alloc(sizeof(int) * k));
    HIPCHK(sc.alloc(sizeof(int) * k));
    HIPCHK(sa.alloc(sizeof(double) * k));
    HIPCHK(sf.alloc(sizeof(float) * k));
    if (nnz > 0) {
        HIPCHK(hipMemcpy(sr.p, irow, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sc.p, icol, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(sf.p, a, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
        const int g = (int)std::min<int64_t>((nnz + VEC_BLOCK - 1) / VEC_BLOCK, 65535);
        hipLaunchKernelGGL((k_convert<float, double>), dim3(g), dim3(VEC_BLOCK), 0, 0, (const float *)sf.as<float>(),
                           sa.as<double>(), nnz);   // exact
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
    }
    H *h = nullptr;
    RET(new_handle(m, n, nnz, &h));
    // all-REAL32 storage on the device unless the mixed mode is asked for (binary64 on the device, real32
    // at the boundary only: LSQRHIP_REAL32_MIXED=1)
    h->f32 = env_int("LSQRHIP_REAL32_MIXED", 0) == 0;
    h->io32 = true;
    int rc = finish_create(h, sr.as<int>(), sc.as<int>(), sa.as<double>());
    if (rc == LSQRHIP_OK) rc = tune_panel_grids(h);
    if (rc != LSQRHIP_OK) {
        std::string keep = g_last_error;
        lsqrhip_destroy(h);
        g_last_error = keep;
        return rc;
    }
    *out = h;
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_solve_f32(lsqrhip_handle_t h, const float *b, double damp, double atol, double btol,
                                 double conlim, int itnlim, int wantse, int want_log, float *x, float *se, int *istop,
                                 int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!h->io32) return fail(LSQRHIP_ERR_ARG, "not a handle of lsqrhip_create_f32");
    if (h->f32)   // float vectors all the way: solve_core moves 4-byte elements for such a handle
        return solve_core(h, reinterpret_cast<const double *>(b), false, damp, atol, btol, conlim, itnlim, wantse, want_log,
                          reinterpret_cast<double *>(x), reinterpret_cast<double *>(se), false, istop, itn, anorm, acond,
                          rnorm, arnorm, xnorm);
    // mixed mode: binary64 on the device, real32 at the boundary
    if ((!b && h->m > 0) || (!x && h->n > 0)) return fail(LSQRHIP_ERR_ARG, "null b or x");
    std::vector<double> bd((size_t)std::max(h->m, 1)), xd((size_t)std::max(h->n, 1)), sd(wantse ? xd.size() : 1);
    for (int i = 0; i < h->m; ++i) bd[(size_t)i] = (double)b[i];
    RET(solve_core(h, bd.data(), false, damp, atol, btol, conlim, itnlim, wantse, want_log, xd.data(),
                   wantse ? sd.data() : nullptr, false, istop, itn, anorm, acond, rnorm, arnorm, xnorm));
    for (int j = 0; j < h->n; ++j) x[j] = (float)xd[(size_t)j];
    if (wantse && se)
        for (int j = 0; j < h->n; ++j) se[j] = (float)sd[(size_t)j];
    return LSQRHIP_OK;
}

extern "C" int lsqrhip_solve_device_f32(lsqrhip_handle_t h, const float *d_b, double damp, double atol, double btol,
                                        double conlim, int itnlim, int wantse, int want_log, float *d_x, float *d_se,
                                        int *istop, int *itn, double *anorm, double *acond, double *rnorm,
                                        double *arnorm, double *xnorm)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!h->f32) return fail(LSQRHIP_ERR_ARG, "not a REAL32 handle (binary64 vectors on the device: lsqrhip_solve_device)");
    return solve_core(h, reinterpret_cast<const double *>(d_b), true, damp, atol, btol, conlim, itnlim, wantse, want_log,
                      reinterpret_cast<double *>(d_x), reinterpret_cast<double *>(d_se), true, istop, itn, anorm, acond,
                      rnorm, arnorm, xnorm);
}

extern "C" int lsqrhip_aprod_device_f32(lsqrhip_handle_t h, int mode, float *d_x, float *d_y)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!h->f32) return fail(LSQRHIP_ERR_ARG, "not a REAL32 handle (binary64 vectors on the device: lsqrhip_aprod_device)");
    h->f32_device_ok = true;
    const int rc = lsqrhip_aprod_device(h, mode, reinterpret_cast<double *>(d_x), reinterpret_cast<double *>(d_y));
    h->f32_device_ok = false;
    return rc;
}

extern "C" int lsqrhip_aprod_f32(lsqrhip_handle_t h, int mode, float *x, float *y)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!h->io32) return fail(LSQRHIP_ERR_ARG, "not a handle of lsqrhip_create_f32");
    if (mode != 1 && mode != 2) return fail(LSQRHIP_ERR_MODE, lsqrhip_error_string(LSQRHIP_ERR_MODE));
    if (h->group && h->f32) return aprod_group_host_f32(h, mode, x, y);   // REAL32 blocks on several devices
    if (!h->f32) {  // mixed mode
        std::vector<double> xd((size_t)std::max(h->n, 1)), yd((size_t)std::max(h->m, 1));
        for (int j = 0; j < h->n; ++j) xd[(size_t)j] = (double)x[j];
        for (int i = 0; i < h->m; ++i) yd[(size_t)i] = (double)y[i];
        RET(lsqrhip_aprod(h, mode, xd.data(), yd.data()));
        if (mode == 1)
            for (int i = 0; i < h->m; ++i) y[i] = (float)yd[(size_t)i];
        else
            for (int j = 0; j < h->n; ++j) x[j] = (float)xd[(size_t)j];
        return LSQRHIP_OK;
    }
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // borrow the solver's work vectors (float arrays): V (n) for x, U (m) for y
    if (h->n > 0) HIPCHK(hipMemcpyAsync(h->V, x, sizeof(float) * (size_t)h->n, hipMemcpyHostToDevice, s));
    if (h->m > 0) HIPCHK(hipMemcpyAsync(h->U, y, sizeof(float) * (size_t)h->m, hipMemcpyHostToDevice, s));
    h->f32_device_ok = true;
    const int rc = lsqrhip_aprod_device(h, mode, h->V, h->U);
    h->f32_device_ok = false;
    RET(rc);
    if (mode == 1 && h->m > 0) HIPCHK(hipMemcpyAsync(y, h->U, sizeof(float) * (size_t)h->m, hipMemcpyDeviceToHost, s));
    if (mode == 2 && h->n > 0) HIPCHK(hipMemcpyAsync(x, h->V, sizeof(float) * (size_t)h->n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return LSQRHIP_OK;
}

// row-block sharded solve (multi-GPU) and on-device problem generators
#include "op_api.h"
#include "shard_api.h"
#include "shard_engine.h"
#include "gen_api.h"
