/*
 * lsqrhip.h -- C-ABI of the MI355X-native LSQR hot path (liblsqrhip.so).
 *
 * This is the drop-in boundary for the reference's `lsqr_solver_ez` path
 * (jacobwilliams/LSQR).  Each entry point names the reference interface it
 * replaces (paths relative to the reference root).  Plain pointers and sizes
 * only; host code in any language binds these (Fortran ISO_C_BINDING shim:
 * lsqr_amd/fortran/lsqr_module.f90; Python ctypes: lsqr_amd/capi.py; see
 * INTEGRATION.md).
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, 1..5 = the reference's own
 *     `error stop` conditions (same ordinal as listed below), >= 10 = runtime
 *     failures (no device, HIP error, allocation).  There is NO CPU fallback:
 *     without a usable gfx950 device every compute entry point fails with
 *     LSQRHIP_ERR_NO_DEVICE.
 *   - COO indices are 1-based (Fortran), vectors are dense fp64.
 *   - pointers named d_* are device pointers; all others are host pointers.
 *   - one handle = one matrix on one GPU with its own HIP stream; calls on one
 *     handle are blocking and must not be issued concurrently (the reference
 *     object carries scratch the same way, src/lsqr.f90:54-58).
 */
#ifndef LSQRHIP_H
#define LSQRHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsqrhip_handle_s *lsqrhip_handle_t;

/* status codes --------------------------------------------------------- */
#define LSQRHIP_OK 0
#define LSQRHIP_ERR_SIZES 1     /* 'invalid a,icol,irow sizes in initialize_ez'   src/lsqr.f90:109 */
#define LSQRHIP_ERR_IROW 2      /* 'invalid irow or m in initialize_ez'           src/lsqr.f90:110 */
#define LSQRHIP_ERR_ICOL 3      /* 'invalid icol or n in initialize_ez'           src/lsqr.f90:111 */
#define LSQRHIP_ERR_NOT_INIT 4  /* 'lsqr_solver_ez class not properly initialized' src/lsqr.f90:152 */
#define LSQRHIP_ERR_MODE 5      /* 'invalid mode in aprod_ez'                     src/lsqr.f90:197 */
#define LSQRHIP_ERR_NO_DEVICE 10
#define LSQRHIP_ERR_HIP 11
#define LSQRHIP_ERR_ALLOC 12
#define LSQRHIP_ERR_ARG 13
#define LSQRHIP_ERR_TOO_LARGE 14

/* The reference's message for codes 1..5 (verbatim `error stop` strings), a
 * short description otherwise. */
const char *lsqrhip_error_string(int code);
/* Detail of the last failure on the calling thread (HIP error text etc.). */
const char *lsqrhip_last_error(void);

/* Number of usable gfx950 devices (0 when none; never initialises a context). */
int lsqrhip_device_count(void);
/* Select the device later handles are created on (default 0). */
int lsqrhip_set_device(int device);

/* ---------------------------------------------------------------------- */
/* initialize_ez                     replaces src/lsqr.f90:91-127           */
/* ---------------------------------------------------------------------- */
/* Validates like the reference (irow > m -> 2, icol > n -> 3; additionally
 * indices < 1, which the reference leaves unchecked and would read out of
 * bounds on, are rejected with the same codes), deep-copies the COO triplets
 * to the device and builds CSR(A) and CSR(A') there (stable: entries of a row
 * keep their COO order, duplicates are kept and therefore summed exactly as
 * src/lsqr.f90:168-172 sums them).  The caller may free its arrays afterwards. */
int lsqrhip_create(int m, int n, int64_t nnz, const int *irow, const int *icol, const double *a,
                   lsqrhip_handle_t *out);

/* Same, from COO triplets that already live on the device (1-based). */
int lsqrhip_create_from_device_coo(int m, int n, int64_t nnz, const int *d_irow, const int *d_icol,
                                   const double *d_a, lsqrhip_handle_t *out);

/* Releases every device resource of the handle (the reference has no finaliser;
 * its allocatables auto-free, src/lsqr.f90:42-58). */
int lsqrhip_destroy(lsqrhip_handle_t h);

/* Add one owner: the handle is freed by the LAST lsqrhip_destroy.  Lets host languages
 * with value semantics (Fortran intrinsic assignment of a solver object, which in the
 * reference deep-copies its allocatable components) share one device matrix safely. */
int lsqrhip_retain(lsqrhip_handle_t h);

/* Matrix facts: dims[0..15] = m, n, nnz, bytes of CSR(A), bytes of CSR(A') as stored,
 * bytes per row pointer (4 or 8), value-dictionary entries (0 = none), bytes per stored
 * value (8, or 1 with the dictionary), bytes per column index in CSR(A) and CSR(A')
 * (4, or 2 for block-relative indices), column panels of CSR(A) and CSR(A'),
 * the short-row layout in use for A and for A' (0 = none, 1 = sliced ELL, 2 = its packed 16-byte records,
 * 3 = row patterns: one byte per row, 4 = structure patterns: a byte per row + the values, csrc/pat.h),
 * whether the panels are LDS-resident
 * for A and for A' (0/1/2; 3 = column-swept row blocks, csrc/csb.h). */
int lsqrhip_info(lsqrhip_handle_t h, int64_t *dims);

/* ---------------------------------------------------------------------- */
/* solve_ez + LSQR                   replaces src/lsqr.f90:207-259, 432-882 */
/* ---------------------------------------------------------------------- */
/* b[m] in, x[n] out, se[n] out iff wantse != 0 (else untouched, may be NULL).
 * atol/btol/conlim/itnlim are the values `initialize` stored (:121-124).
 * want_log != 0 keeps the per-iteration record the reference would print for
 * nout /= 0 (:813-837); fetch it with lsqrhip_log_fetch.  Any scalar output
 * pointer may be NULL (the Fortran `optional` outputs, :217-223).
 * Deliberate fix: when no iteration runs (istop = 0) rnorm is returned as
 * norm(b); the reference leaves it unassigned (:646-653). */
int lsqrhip_solve(lsqrhip_handle_t h, const double *b, double damp, double atol, double btol,
                  double conlim, int itnlim, int wantse, int want_log, double *x, double *se,
                  int *istop, int *itn, double *anorm, double *acond, double *rnorm,
                  double *arnorm, double *xnorm);

/* Same with b, x, se resident in HBM (no PCIe in the call). */
int lsqrhip_solve_device(lsqrhip_handle_t h, const double *d_b, double damp, double atol,
                         double btol, double conlim, int itnlim, int wantse, int want_log,
                         double *d_x, double *d_se, int *istop, int *itn, double *anorm,
                         double *acond, double *rnorm, double *arnorm, double *xnorm);

/* ---------------------------------------------------------------------- */
/* REAL32: the reference's precision macro   src/lsqr_kinds.F90:16-17 (wp = real32) */
/* ---------------------------------------------------------------------- */
/* The same three entry points for a host built with wp = real32 (lsqr_amd/fortran with -DREAL32 binds
 * these).  Storage on the device is real32 END TO END -- the matrix values and every vector of the
 * iteration (u, v, w, x, se): half the bytes of every vector pass and of the value stream -- while the
 * arithmetic in registers stays binary64 (products, row sums, norms, the scalar recurrences), so the
 * result is at least as accurate as the reference's all-real32 iteration.  The tolerances, damp and the
 * returned scalars are doubles (exact conversions of the host's real32 values).
 * LSQRHIP_REAL32_MIXED=1 (environment, read at create): the mixed mode -- binary64 storage on the device,
 * real32 only at this boundary.
 * Layouts: row and structure patterns, sliced ELL, row windows, column-swept row blocks (no panel kernels).  A REAL32 handle works
 * with lsqrhip_solve_f32, lsqrhip_aprod_f32, the log, timing, option and info entry points; the binary64
 * entry points refuse it. */
int lsqrhip_create_f32(int m, int n, int64_t nnz, const int *irow, const int *icol, const float *a,
                       lsqrhip_handle_t *out);
int lsqrhip_solve_f32(lsqrhip_handle_t h, const float *b, double damp, double atol, double btol, double conlim,
                      int itnlim, int wantse, int want_log, float *x, float *se, int *istop, int *itn,
                      double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm);
int lsqrhip_aprod_f32(lsqrhip_handle_t h, int mode, float *x, float *y);
/* b, x, se (and the aprod vectors) already in HBM as real32 arrays; all-real32 handles only (a mixed-mode
 * handle keeps binary64 vectors on the device: lsqrhip_solve_device / lsqrhip_aprod_device). */
int lsqrhip_solve_device_f32(lsqrhip_handle_t h, const float *d_b, double damp, double atol, double btol,
                             double conlim, int itnlim, int wantse, int want_log, float *d_x, float *d_se,
                             int *istop, int *itn, double *anorm, double *acond, double *rnorm, double *arnorm,
                             double *xnorm);
int lsqrhip_aprod_device_f32(lsqrhip_handle_t h, int mode, float *d_x, float *d_y);

/* ---------------------------------------------------------------------- */
/* aprod_ez                          replaces src/lsqr.f90:134-200          */
/* ---------------------------------------------------------------------- */
/* mode 1: y[m] += A x[n] (x unchanged); mode 2: x[n] += A' y[m] (y unchanged). */
int lsqrhip_aprod(lsqrhip_handle_t h, int mode, double *x, double *y);
int lsqrhip_aprod_device(lsqrhip_handle_t h, int mode, double *d_x, double *d_y);

/* ---------------------------------------------------------------------- */
/* acheck / xcheck on the device operator   replaces src/lsqr.f90:908-994, 1015-1154 */
/* ---------------------------------------------------------------------- */
int lsqrhip_acheck(lsqrhip_handle_t h, double eps, int *inform, double *relerr);
/* u[m], v[n], w[n] receive r = b - Ax, A'r, A'r - damp^2 x (may be NULL). tests[3]. */
int lsqrhip_xcheck(lsqrhip_handle_t h, double anorm, double damp, double eps, const double *b,
                   const double *x, double *u, double *v, double *w, int *inform, double *tests);
/* ... for REAL32 handles (matrix handles of lsqrhip_create_f32, operator handles of lsqrhip_create_operator_f32):
 * real32 vectors on the device and at the boundary, binary64 arithmetic between them (src/lsqr.f90:908-994,
 * 1015-1154 under src/lsqr_kinds.F90:16-17). */
int lsqrhip_acheck_f32(lsqrhip_handle_t h, double eps, int *inform, double *relerr);
int lsqrhip_xcheck_f32(lsqrhip_handle_t h, double anorm, double damp, double eps, const float *b, const float *x,
                       float *u, float *v, float *w, int *inform, double *tests);

/* ---------------------------------------------------------------------- */
/* iteration log (nout /= 0)         replaces src/lsqr.f90:813-837          */
/* ---------------------------------------------------------------------- */
/* One record per PRINTED iteration (the device applies the reference's selective print rule,
 * :815-822, so the buffer stays small for huge itnlim), LSQRHIP_LOG_STRIDE doubles: itn, x(1), rnorm,
 * test1, test2, anorm, acond, phi, dknorm, dxk, alfa_opt, istop (as decided in
 * that iteration before the nconv rule), rtol, xnorm.  Host code applies the
 * reference's print rule (:815-822) and format strings to them. */
#define LSQRHIP_LOG_STRIDE 14
int lsqrhip_log_count(lsqrhip_handle_t h);
int lsqrhip_log_fetch(lsqrhip_handle_t h, int first, int count, double *records);
/* Scalars of the last solve that only the log prints: out[0..5] = bnorm, dxmax,
 * maxdx, alpha(first), beta(first), test2(itn 0) (src/lsqr.f90:663-669, 875-878). */
int lsqrhip_log_extras(lsqrhip_handle_t h, double *out);

/* ---------------------------------------------------------------------- */
/* device BLAS-1 used inside the iteration   replaces src/lsqrblas.f90:25-201 */
/* ---------------------------------------------------------------------- */
/* Unit-stride, device vectors; results returned to the host. */
int lsqrhip_dnrm2(lsqrhip_handle_t h, int64_t n, const double *d_x, double *result);
int lsqrhip_ddot(lsqrhip_handle_t h, int64_t n, const double *d_x, const double *d_y, double *result);
int lsqrhip_dscal(lsqrhip_handle_t h, int64_t n, double da, double *d_x);
int lsqrhip_dcopy(lsqrhip_handle_t h, int64_t n, const double *d_x, double *d_y);

/* ---------------------------------------------------------------------- */
/* measurement                                                             */
/* ---------------------------------------------------------------------- */
typedef struct {
    double solve_ms;      /* whole lsqrhip_solve* call, host clock                     */
    double loop_ms;       /* device time of the iteration loop (HIP events): option
                             "loop_events" or "time_kernels", -1 otherwise              */
    double spmv1_ms;      /* summed device time of the mode-1 SpMV kernel launches     */
    double spmv2_ms;      /* summed device time of the mode-2 SpMV kernel launches     */
    double update_ms;     /* summed device time of the x/w update kernel launches      */
    int64_t spmv1_launches, spmv2_launches, update_launches;
    int64_t spmv1_bytes;  /* algorithmic bytes of ONE mode-1 launch (B1) */
    int64_t spmv2_bytes;  /* algorithmic bytes of ONE mode-2 launch (B2)               */
    int64_t vec_bytes;    /* algorithmic bytes of the fused vector work per iteration  */
    int itn;
} lsqrhip_timing_t;
int lsqrhip_last_timing(lsqrhip_handle_t h, lsqrhip_timing_t *t);

/* Average device time (ms) of `reps` back-to-back launches of one hot kernel on the
 * handle's stream, bracketed by ONE pair of HIP events (so the per-launch figure is what
 * rocprofv3's kernel trace reports as the kernel's average duration).  which: 1 = mode-1
 * SpMV on CSR(A), 2 = mode-2 SpMV on CSR(A'), 3 = x/w update.  Runs on scratch
 * copies of the coefficients with the solver's work vectors as operands; their contents
 * are unspecified afterwards. */
int lsqrhip_bench_kernel(lsqrhip_handle_t h, int which, int reps, double *avg_ms);

/* ---- LSQR on a user-supplied DEVICE operator ---------------------------------
 * Replaces the reference's abstract class `lsqr_solver` with a deferred `aprod`
 * (src/lsqr.f90:16-30; interface :67-82) for operators that live on the GPU.  The callback
 * must ENQUEUE on `hip_stream` (a hipStream_t)
 *     mode 1:  y <- y + A  x          mode 2:  x <- x + A' y
 * with x (n) and y (m) DEVICE pointers, and return 0.  It must not synchronise with, or wait
 * for, the host: the solver queues several iterations ahead.  Past the stopping iteration it
 * may still be called, with an all-zero input vector.
 * The handle works with lsqrhip_solve / _solve_device (LSQR, :432-882), lsqrhip_aprod /
 * _aprod_device, lsqrhip_acheck (:908-994), lsqrhip_xcheck (:1015-1154), the log and BLAS-1
 * entry points; option "op_batch" = iterations queued ahead of each stop poll (default 8). */
typedef int (*lsqrhip_aprod_fn)(void *user, int mode, int m, int n, double *d_x, double *d_y, void *hip_stream);
int lsqrhip_create_operator(int m, int n, lsqrhip_aprod_fn aprod, void *user, lsqrhip_handle_t *h);
/* The same in the reference's REAL32 build (src/lsqr_kinds.F90:16-17 makes wp = real32 for the abstract class too,
 * src/lsqr.f90:16-30): x, y and every work vector of the iteration are real32 arrays on the device; arithmetic in
 * registers stays binary64.  Solve with lsqrhip_solve_f32 / lsqrhip_solve_device_f32, apply with lsqrhip_aprod_f32 /
 * lsqrhip_aprod_device_f32, check with lsqrhip_acheck_f32 / lsqrhip_xcheck_f32. */
typedef int (*lsqrhip_aprod_f32_fn)(void *user, int mode, int m, int n, float *d_x, float *d_y, void *hip_stream);
int lsqrhip_create_operator_f32(int m, int n, lsqrhip_aprod_f32_fn aprod, void *user, lsqrhip_handle_t *h);

/* The reference's own test operator A = HY * D * HZ as a device operator, with the problem
 * generator `lstp` (test/lsqrtest_module.f90: hprod :385-403, aprod1 :319-343, aprod2 :353-377,
 * lstp :422-505): creates an operator handle for problem P(m, n, nduplc, npower, damp) and
 * returns the generator's condition number and residual norm.  lsqrhip_lstp_vectors copies out
 * the generated xtrue (n), b (m), d (min(m,n)), hy (m), hz (n) (any may be NULL) and the device
 * address of b. */
int lsqrhip_lstp_create(int m, int n, int nduplc, int npower, double damp, lsqrhip_handle_t *h, double *acond,
                        double *rnorm);
int lsqrhip_lstp_vectors(lsqrhip_handle_t h, double *xtrue, double *b, double *d, double *hy, double *hz,
                         const double **d_b);
/* ... generated in binary32 arithmetic, as the reference's test module is under -DREAL32, and held in real32 arrays on
 * the device (an operator handle of the REAL32 kind; lsqrhip_lstp_vectors returns its vectors widened to double, d_b
 * points at a float array). */
int lsqrhip_lstp_create_f32(int m, int n, int nduplc, int npower, double damp, lsqrhip_handle_t *h, double *acond,
                            double *rnorm);

/* Options: "graph" (1 = hipGraph-captured iteration batches [default], 0 = eager
 * launches), "graph_iters" (iterations per captured batch, default 64), "time_kernels"
 * (1 = eager launches with HIP events around each hot kernel, fills *_ms above),
 * "pipeline" (launch schedule of the loop; all three give the same bits:
 * 0 = K1 S1 K2 S2 K4 S3, 1 = scalar steps ride inside the SpMV launches,
 * 2 = additionally the x/w update rides inside the mode-1 launch [default]),
 * "poll_ahead" (1 [default] = with graphs, the next batch is enqueued before the host waits
 * for the current one's stop flag, so the poll and the graph launch overlap device work; a
 * solve that stops inside batch k then runs batch k+1 as no-op launches; 0 = strict
 * launch-wait-check), "loop_events" (1 = two HIP events around the iteration loop fill
 * timing.loop_ms; default 0: the records cost a short solve 8-9 us). */
int lsqrhip_set_option(lsqrhip_handle_t h, const char *name, int64_t value);
/* Reads an option back.  Besides the above: "norm_exp" -- the fused in-loop norms (dnrm2,
 * src/lsqrblas.f90:123-159) are sqrt(sum (y 2^-e)^2) 2^e with e = norm_exp fixed per matrix
 * (2^e just above max|a_ij|: u and v live at the scale of the matrix), so that no norm over- or
 * underflows whatever the scale of A and b.  The ranks of a row-sharded solve must agree on one e
 * (set the maximum of their values on every rank before lsqrhip_shard_begin).
 * "log_truncated": 1 when the last solve had more printable iterations than the log buffer holds
 * (itnlim / 10 + 64 records for n > 40): the earliest overflowing records were dropped, the record of
 * the stopping iteration is always kept (last slot).  "launches_mode1" / "launches_mode2": kernel
 * launches one product takes in the layout in use (for reading profiler output).
 * Round 5: "csb_lockstep_mode1" / "_mode2" -- chunks per wave and lock-step step of the column-swept product in use
 * (0: the free-running sweep, or another layout); "shard_overlap", "shard_parts", "shard_copy" -- the
 * schedule the sharded engine of this handle REALLY runs (a requested overlap or copy mode that could not be set up
 * reads 0); "shard_engine_flags" -- what the C++ engine left in this rank's stage context (0 between solves, also
 * after an engine solve that failed half way: 1 own slice read in T | 2 long norms message | 4 norms gathered by the
 * engine | 8 a sharded solve is open); "pat_wide_mode1" / "_mode2" -- distinct rows of the wide row-pattern table
 * in use (two-byte pattern numbers; 0: the one-byte table or another layout); "pat_pair_mode1" / "_mode2" -- 1 when the
 * row-pattern product runs in paired rows (a lane owns two neighbouring rows: DESIGN.md 3.4b).
 * Round 6: "csb_fuse_mode1" / "_mode2" -- 1 when a column-split product is closed by its last arriving split inside the
 * sweep launch (no k_csb_combine launch: the build's choice for two splits per block; LSQRHIP_CSB_FUSE=0 / 1 forces); "csb_probe_mode1" /
 * "_mode2" -- device address of the product's phase clocks when the handle was created with LSQRHIP_CSB_PROBE=1
 * (measurement only: scripts/csb_probe.py), else 0. */
int lsqrhip_get_option(lsqrhip_handle_t h, const char *name, int64_t *value);
/* Run all work of this handle on an externally owned hipStream_t (e.g. the
 * caller's torch stream); NULL restores the handle's own stream. */
int lsqrhip_set_stream(lsqrhip_handle_t h, void *hip_stream);

/* plain device-memory helpers for hosts without their own allocator */
int lsqrhip_dev_alloc(void **d_ptr, int64_t bytes);
int lsqrhip_dev_free(void *d_ptr);
int lsqrhip_dev_upload(void *d_dst, const void *src, int64_t bytes);
int lsqrhip_dev_download(void *dst, const void *d_src, int64_t bytes);
int lsqrhip_dev_sync(void);

/* ---------------------------------------------------------------------- */
/* row-block sharded solve over the GPUs of one node                        */
/* ---------------------------------------------------------------------- */
/* The reference is serial; this is the multi-GPU form of the same iteration:
 * A = [A_1; ...; A_P] in contiguous row blocks, u and b sharded with the rows, v replicated,
 * x / w / se sharded by column slices of chunk = ceil(n / P); per iteration an all-reduce of one
 * double (beta), a direct reduce-scatter of the n-vector A'u (every slice summed in rank order by
 * its owner), an all-reduce of two doubles (alpha, dknorm) and an all-gather of the v slices.
 *
 * (1) ONE process, ngpu devices -- what `initialize(..., ngpu=N)` of the Fortran layer binds
 *     (replaces initialize_ez, src/lsqr.f90:91-127, for a node's worth of GPUs).  Same arguments and
 *     validation as lsqrhip_create; rows are cut into ngpu blocks balanced by nonzeros, one sub-handle
 *     per device (devices d0 .. d0+ngpu-1, d0 = lsqrhip_set_device's), RCCL communicator inside.  The
 *     handle then works with lsqrhip_solve (host b, x, se: replaces solve_ez :207-259 + LSQR :432-882),
 *     lsqrhip_aprod (host vectors, :134-200), lsqrhip_info, lsqrhip_retain, lsqrhip_destroy.
 *     ngpu is clamped to the number of rows; more devices than the node has -> LSQRHIP_ERR_NO_DEVICE. */
int lsqrhip_create_sharded(int m, int n, int64_t nnz, const int *irow, const int *icol, const double *a, int ngpu,
                           lsqrhip_handle_t *out);
/*     The same for a host built with wp = real32 (src/lsqr_kinds.F90:16-17; lsqr_amd/fortran -DREAL32 with ngpu=):
 *     real32 storage in every row block and real32 slices on the links (half the bytes of both exchanges), binary64
 *     registers; the handle then works with lsqrhip_solve_f32 / lsqrhip_aprod_f32 (float host vectors).
 *     LSQRHIP_REAL32_MIXED=1: binary64 on the devices, real32 at the boundary only. */
int lsqrhip_create_sharded_f32(int m, int n, int64_t nnz, const int *irow, const int *icol, const float *a, int ngpu,
                               lsqrhip_handle_t *out);

/* (2) one process PER GPU (bench.py --gpus N under torch.distributed.run): every rank creates an
 *     ordinary handle from its own row block (rows renumbered from 1, all n columns), rank 0 obtains a
 *     128-byte RCCL id and the host program hands it to the others (any transport), every rank joins,
 *     and lsqrhip_shard_solve runs the whole loop -- kernels and RCCL calls -- from C++:
 *     d_b_local (m_p) in, d_x (n, and d_se if wantse) out on every rank, all device pointers.
 *     Environment of EVERY rank, read by lsqrhip_shard_comm_init: LSQRHIP_SHARD_OVERLAP=1 (with LSQRHIP_SHARD_WORLD at the
 *     handle's create: exchanges in parts beside the products), LSQRHIP_SHARD_COPY=1 (round 5: the n-vector exchanges as
 *     copy-engine pulls from the peers' buffers, mapped by hipIpcOpenMemHandle -- no RCCL send / receive kernel; both may
 *     be set); lsqrhip_get_option "shard_overlap" / "shard_copy" report what took. */
int lsqrhip_rccl_unique_id(char *out128);
int lsqrhip_shard_comm_init(lsqrhip_handle_t h, int world, int rank, int64_t row0, int64_t m_global,
                            const char *id128);
int lsqrhip_shard_solve(lsqrhip_handle_t h, const double *d_b_local, double damp, double atol, double btol,
                        double conlim, int itnlim, int wantse, double *d_x, double *d_se, int *istop, int *itn,
                        double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm);

/* (3) the stages themselves, for a host that carries the exchanges (lsqr_amd/dist.py over
 *     torch.distributed; gloo in the CPU tests).  The library launches only LOCAL kernels, asynchronously
 *     on the handle's stream; between the stages the caller exchanges buffers it owns:
 *
 *   begin(b_p, world, rank, T, R, V, sums)       T, R, V: world * chunk doubles, sums: 4 doubles
 *   stage 0                 -> all-reduce sums[0..2]                         three range-safe sums of b^2
 *   stage 1                 -> R[r * chunk ..] <- rank r's T[rank * chunk ..]  for every r (all-to-all)
 *   stage 2                 -> all-reduce sums[0..1]
 *   stage 3                 -> all-gather V (slice of rank q at q * chunk)
 *   repeat:  stage 4        -> all-reduce sums[0]                            |u|^2
 *            stage 5        -> the all-to-all T -> R                         A'u
 *            stage 6        -> all-reduce sums[0..1]                         |v|^2, |w|^2
 *            stage 7        -> all-gather V
 *            every k iterations: poll (all ranks see the same stop flag: the scalar recurrences run
 *            replicated on identical all-reduced inputs)
 *   end(x, se)              this rank's slices land at [rank * chunk, ..) of x, se: all-gather them
 *
 * All-reduces of the scalars must give every rank the SAME bits (shard_engine.h and dist.py sum the
 * ranks' values in rank order). */
int lsqrhip_shard_begin(lsqrhip_handle_t h, const double *d_b_local, int64_t m_global, int world, int rank,
                        double damp, double atol, double btol, double conlim, int itnlim, int wantse, double *d_T,
                        double *d_R, double *d_V, double *d_sums);
int lsqrhip_shard_stage(lsqrhip_handle_t h, int stage);
/* d_out[0..chunk) = sum_{r < nchunks} d_in[r*chunk + i], in rank order (asynchronous on the
 * handle's stream). */
int lsqrhip_sum_chunks(lsqrhip_handle_t h, const double *d_in, int nchunks, int64_t chunk, double *d_out);
/* out[0..2] = stop, itn, istop (synchronises the handle's stream). */
int lsqrhip_shard_poll(lsqrhip_handle_t h, int *out);
int lsqrhip_shard_end(lsqrhip_handle_t h, double *d_x, double *d_se, int *istop, int *itn, double *anorm,
                      double *acond, double *rnorm, double *arnorm, double *xnorm);

/* ---------------------------------------------------------------------- */
/* synthetic systems generated in HBM (bench / scale tests)                 */
/* ---------------------------------------------------------------------- */
/* Bit-identical to the host generators in lsqr_amd/problems.py (same counter-based hash).
 * kind 0: random rows (p0 = nnz per row); kind 1: 5-point Poisson (p0 = nx, p1 = ny; b is
 * not generated, d_b may be NULL); kind 2: rows with prescribed degrees (power law):
 * d_rowptr_local = exclusive prefix of the local row degrees, nrows+1 entries, built by the
 * host from the integer CDF table (problems.powerlaw_degrees); kind 3: a five-point mesh whose
 * coefficient is constant on each of bx x by regions (problems.mesh2d: p0 = nx, ny = m / nx,
 * p1 = bx << 16 | by).
 * Generates rows [row0, row0+nrows) of the global m-by-n system as LOCAL 1-based COO
 * (irow in 1..nrows) plus b for those rows.  *nnz_out = triplets written. */
int64_t lsqrhip_gen_count(int kind, int64_t m, int64_t n, int64_t p0, int64_t p1, int64_t row0, int64_t nrows);
int lsqrhip_gen_coo(int kind, uint64_t seed, int64_t m, int64_t n, int64_t p0, int64_t p1, int64_t row0,
                    int64_t nrows, const int64_t *d_rowptr_local, int *d_irow, int *d_icol, double *d_a,
                    double *d_b, int64_t *nnz_out);

#ifdef __cplusplus
}
#endif
#endif /* LSQRHIP_H */
