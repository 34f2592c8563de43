// shard_api.h -- row-block sharded LSQR: the per-rank stages (included by lsqrhip.hip).
//
// Rank p of P holds A_p = rows [row0, row0 + m_p) of A (its own handle: the layouts of A_p and A_p').
//     u, b          sharded with the rows                     (m_p entries)
//     v             replicated: mode 1 gathers from all of it (P * chunk entries, chunk = ceil(n / P))
//     x, w, se      sharded by COLUMN slices: rank q owns columns [q chunk, (q+1) chunk)
// Per iteration (the "better" form):
//     u_p  <- A_p v - alpha u_p                                      local
//     sum_p |u_p|^2                                                  all-reduce, 1 double        (beta)
//     T_p  =  A_p' u_p                                               local, n doubles
//     reduce-scatter of T: slice q of every T_p goes to rank q       all-to-all over every xGMI link at once,
//        and is summed there in RANK ORDER (deterministic)           (P-1)/P * 8n bytes out and in per GPU
//     v_q  <- T_q - beta v_q ;  |v_q|^2, and |w_q|^2 of the last update        all-reduce, 2 doubles (alpha, dknorm)
//     rotations; x_q += t1 w_q ; w_q <- t2 w_q + v_q ; stopping tests          slice-local (n / P each)
//     all-gather of the v slices                                     (P-1)/P * 8n bytes in per GPU
// so the n-vector work scales with P like the products do, and the only replicated work is the
// scalar recurrences (bit-identical on every rank: their inputs are all-reduced values).
// dknorm = sqrt(sum (t3 w_i)^2) (src/lsqr.f90:729-745) is formed as |t3| sqrt(sum w_i^2), so that its
// sum rides in the same all-reduce as |v|^2 (w is the PREVIOUS iteration's, known before t3 is).
//
// This file only launches local kernels on the handle's stream, asynchronously.  The collectives
// are issued by the caller between the stages on buffers it owns (T, R, V, sums): by
// shard_engine.h (RCCL, from C++: lsqrhip_create_sharded / lsqrhip_shard_solve), or by
// lsqr_amd/dist.py (torch.distributed; gloo in the CPU tests, which pin the arithmetic).
#pragma once

namespace lsqrhip {

// Vq_i <- cy (Vq_i sy) + (R[0][i] + R[1][i] + ... + R[P-1][i])   (rank order); partials of sum (Vq ns)^2.
// R[r] = slice q of rank r's T (what the all-to-all delivered), each `chunk` long; i < len <= chunk.
// `own` (or null: all P slices lie in R): this rank's own slice, read where mode 2 left it in T -- it never travels.
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_rs_combine(VT *__restrict__ Vq, const VT *__restrict__ R,
                                                          const VT *__restrict__ own, int rank,
                                                          int P, int64_t chunk, int64_t len,
                                                          const SpmvCoef *__restrict__ coef,
                                                          const int *__restrict__ stop,
                                                          double *__restrict__ partials, NScale nsc,
                                                          double *__restrict__ maxpart)
{   // maxpart (or null): [gridDim.x] max |Vq_i| of this workgroup's strided share -- the piece maxima csb.h wants of
    // the vector the next mode-1 product gathers from, taken where the slice is written instead of in a pass of its own
    if (*stop != 0 || coef->skip != 0) return;
    const double sy = coef->sy, cy = coef->cy;
    __shared__ double red[VEC_BLOCK / WAVE];
    double s = 0.0, mx = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < len; i += stride) {
        double t = (own != nullptr && rank == 0) ? (double)own[i] : (double)R[i];
        for (int r = 1; r < P; ++r)
            t = t + ((own != nullptr && r == rank) ? (double)own[i] : (double)R[(int64_t)r * chunk + i]);
        const VT v = (VT)(cy * ((double)Vq[i] * sy) + t);
        Vq[i] = v;
        const double vs = (double)v * nsc.s;
        s += vs * vs;
        mx = fmax(mx, fabs((double)v));
    }
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    if (maxpart != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, WAVE));
        __syncthreads();
        if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            double m = red[0];
            for (int i = 1; i < VEC_BLOCK / WAVE; ++i) m = fmax(m, red[i]);
            maxpart[blockIdx.x] = m;
        }
    }
}

// out[i] = in[0*chunk + i] + in[1*chunk + i] + ... in rank order (for callers that reduce a whole
// vector themselves: lsqrhip_sum_chunks)
__global__ __launch_bounds__(VEC_BLOCK) void k_sum_chunks(double *__restrict__ out, const double *__restrict__ in,
                                                          int nchunks, int64_t chunk)
{
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < chunk; i += stride) {
        double s = in[i];
        for (int r = 1; r < nchunks; ++r) s = s + in[(int64_t)r * chunk + i];
        out[i] = s;
    }
}

// The x / w / se update on this rank's column slice (src/lsqr.f90:729-745 with the dscal of :697
// folded in), gated by `live` (set by k_shard_s2 in the SAME iteration: the stop flag that k_shard_s3
// raises afterwards must not hide the last update).  partials = sum of w_new^2 (next iteration's dknorm).
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_update_slice(VT *__restrict__ x, VT *__restrict__ w,
                                                            const VT *__restrict__ Vq, VT *__restrict__ se,
                                                            int64_t len, const LsqrState *__restrict__ st,
                                                            const int *__restrict__ live,
                                                            double *__restrict__ partials)
{
    if (*live == 0) return;
    const double t1 = st->t1, t2 = st->t2, t3 = st->t3, sv = st->sv;
    const bool wantse = st->wantse != 0;
    __shared__ double red[VEC_BLOCK / WAVE];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < len; i += stride) {
        const double t = (double)w[i];
        x[i] = (VT)(t1 * t + (double)x[i]);
        const VT wn = (VT)(t2 * t + (double)Vq[i] * sv);
        w[i] = wn;
        if (wantse) {
            const double d = (t3 * t) * (t3 * t);
            se[i] = (VT)(d + (double)se[i]);
        }
        s += (double)wn * (double)wn;
    }
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// w_q <- V_q sv (first w, src/lsqr.f90:641-644); partials of sum w_q^2
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_init_w_slice(VT *__restrict__ w, const VT *__restrict__ Vq,
                                                            int64_t len, const LsqrState *__restrict__ st,
                                                            double *__restrict__ partials)
{
    const double sv = st->sv;
    __shared__ double red[VEC_BLOCK / WAVE];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    if (st->stop == 0)
        for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < len; i += stride) {
            const VT wn = (VT)((double)Vq[i] * sv);
            w[i] = wn;
            s += (double)wn * (double)wn;
        }
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// sums[0] <- the np partials of sum (V_q ns)^2 in fixed order, sums[1] <- *wsq (this rank's sum of w_q^2)
// ... and, `maxpart` given (the C++ engine: its message is SHARD_MSG doubles), sums[4 + i] <- the largest of the
// workgroup maxima i, i + SHARD_NMAX, ... : this rank's SHARD_NMAX piece maxima of |V_q| (k_rs_combine).  A skipped
// product left V_q as it was: its maxima of the last product stand.
constexpr int SHARD_MSG = 64;                 // doubles a rank contributes to the exchange of the norms
constexpr int SHARD_NMAX = SHARD_MSG - 4;     // ... of which piece maxima of its slice of v
static_assert(VEC_BLOCK == 256 && SHARD_NMAX <= 64, "k_shard_sums: four 64-thread parts, one piece per lane");
__global__ __launch_bounds__(VEC_BLOCK) void k_shard_sums(const double *__restrict__ partials, int np,
                                                          const double *__restrict__ wsq, double *__restrict__ sums,
                                                          const SpmvCoef *__restrict__ coef,
                                                          const double *__restrict__ maxpart)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    const double s = np > 0 ? strided_sum<VEC_BLOCK>(partials, np) : 0.0;
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) {
        sums[0] = coef->skip != 0 ? 0.0 : tot;  // a skipped product left no partials (beta == 0, :691)
        sums[1] = *wsq;
    }
    if (maxpart != nullptr && coef->skip == 0) {   // piece i: workgroups i, i + NMAX, ... -- four threads share one piece
        __shared__ double mx[4][SHARD_NMAX];
        const int piece = threadIdx.x % 64, part = threadIdx.x / 64;   // (VEC_BLOCK = 256: parts 0..3)
        double m = 0.0;
        if (piece < SHARD_NMAX)
            for (int i = piece + part * SHARD_NMAX; i < np; i += 4 * SHARD_NMAX) m = fmax(m, maxpart[i]);
        __syncthreads();
        if (piece < SHARD_NMAX) mx[part][piece] = m;
        __syncthreads();
        if ((int)threadIdx.x < SHARD_NMAX)
            sums[4 + threadIdx.x] = fmax(fmax(mx[0][threadIdx.x], mx[1][threadIdx.x]), fmax(mx[2][threadIdx.x], mx[3][threadIdx.x]));
    }
}

// step 2 on all-reduced sums: alpha = sqrt(sums[0]) / ns, rotations, t1..t3; *live = "this iteration runs"
__global__ void k_shard_s2(const double *__restrict__ sums, LsqrState *st, int *__restrict__ live)
{
    if (st->stop != 0) {
        *live = 0;
        return;
    }
    *live = 1;
    s2_step(st, sqrt(sums[0]) * st->ns_inv, st->c2.skip != 0);
}

// step 3 on all-reduced sums: dknorm^2 = t3^2 * sum of the previous w^2
__global__ void k_shard_s3(const double *__restrict__ sums, LsqrState *st, const int *__restrict__ live,
                           const void *__restrict__ x, int f32, double *__restrict__ log)
{
    if (*live == 0) return;
    const double x1 = x == nullptr ? 0.0 : (f32 ? (double)static_cast<const float *>(x)[0] : static_cast<const double *>(x)[0]);
    s3_step(st, (st->t3 * st->t3) * sums[1], x1, log);
}

// ---- the C++ engine's forms of the three scalar steps (shard_engine.h): each takes over the small kernel that used
// to run in front of it -- the rank-ordered sum of the gathered norms (k_sum_ranks) or the reduction of this rank's
// partials of sum w^2 -- so that an iteration has three launches fewer.  `src`: the P messages of `msg` doubles the
// all-gather delivered (a world of one: the rank's own message).  Same sums in the same order: the same bits.
__global__ void k_shard_s1g(const double *__restrict__ src, int P, int msg, LsqrState *st)
{
    if (st->stop != 0 || threadIdx.x != 0) return;
    double s = src[0];
    for (int r = 1; r < P; ++r) s = s + src[msg * r];
    s1_step(st, sqrt(s) * st->ns_inv);
}

__global__ __launch_bounds__(64) void k_shard_s2g(const double *__restrict__ src, int P, int msg,
                                                  double *__restrict__ sums, LsqrState *st, int *__restrict__ live,
                                                  double *__restrict__ vmax)
{
    if (vmax != nullptr && msg == SHARD_MSG)   // the ranks' piece maxima of |v| side by side (k_sum_ranks)
        for (int i = threadIdx.x; i < P * SHARD_NMAX; i += blockDim.x)
            vmax[i] = src[(i / SHARD_NMAX) * SHARD_MSG + 4 + i % SHARD_NMAX];
    if (threadIdx.x != 0) return;
    double s0 = src[0], s1 = src[1];
    for (int r = 1; r < P; ++r) {
        s0 = s0 + src[msg * r];
        s1 = s1 + src[msg * r + 1];
    }
    sums[0] = s0;
    sums[1] = s1;   // (step 3 reads it after the update)
    if (st->stop != 0) {
        *live = 0;
        return;
    }
    *live = 1;
    s2_step(st, sqrt(s0) * st->ns_inv, st->c2.skip != 0);
}

// k_shard_s2g + k_update_slice in ONE launch (round 5: an engine iteration has a launch fewer).  EVERY workgroup sums the
// ranks' gathered norms in rank order and evaluates the rotation itself (scalar.h rot_step: a pure function) from inputs
// that no workgroup of this launch writes -- alpha from the sums (or, product skipped, the alpha and sv step 2 then leaves
// alone), beta, rhobar / phibar of the OTHER parity -- exactly as the one-handle loop's fused update does; workgroup 0 is
// also the scalar machine (what k_shard_s2g was): sums[0..1], `live`, the state, the ranks' piece maxima side by side.
// Same functions on the same inputs: t1, t2, t3, sv are the bits s2_step stores.
template <typename VT>
__global__ __launch_bounds__(VEC_BLOCK) void k_update_slice_g(VT *__restrict__ x, VT *__restrict__ w,
                                                              const VT *__restrict__ Vq, VT *__restrict__ se, int64_t len,
                                                              LsqrState *st, int *__restrict__ live,
                                                              double *__restrict__ partials,
                                                              const double *__restrict__ src, int P, int msg,
                                                              double *__restrict__ sums, double *__restrict__ vmax)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    __shared__ double sh[8];
    if (threadIdx.x == 0) {
        double s0 = src[0], s1 = src[1];
        for (int r = 1; r < P; ++r) {
            s0 = s0 + src[msg * r];
            s1 = s1 + src[msg * r + 1];
        }
        const bool stopped = st->stop != 0;
        const bool skipped = st->c2.skip != 0;
        const double alpha = skipped ? st->alpha : sqrt(s0) * st->ns_inv;
        const double sv = skipped ? st->sv : (alpha > 0.0 ? 1.0 / alpha : 1.0);
        const int k = st->itn & 1;
        const Rot r = rot_step(st->rhobar2[k ^ 1], st->phibar2[k ^ 1], st->damp, st->damped, alpha, st->beta);
        sh[0] = r.t1;
        sh[1] = r.t2;
        sh[2] = r.t3;
        sh[3] = sv;
        sh[4] = stopped ? 0.0 : 1.0;
        sh[5] = s0;
        sh[6] = s1;
        sh[7] = st->wantse != 0 ? 1.0 : 0.0;
    }
    __syncthreads();
    const bool on = sh[4] != 0.0;
    if (blockIdx.x == 0) {   // the scalar machine
        if (vmax != nullptr && msg == SHARD_MSG)
            for (int i = threadIdx.x; i < P * SHARD_NMAX; i += VEC_BLOCK)
                vmax[i] = src[(i / SHARD_NMAX) * SHARD_MSG + 4 + i % SHARD_NMAX];
        if (threadIdx.x == 0) {
            sums[0] = sh[5];
            sums[1] = sh[6];   // (step 3 reads it after the update)
            *live = on ? 1 : 0;
            if (on) s2_step(st, sqrt(sh[5]) * st->ns_inv, st->c2.skip != 0);
        }
    }
    if (!on) return;
    const double t1 = sh[0], t2 = sh[1], t3 = sh[2], sv = sh[3];
    const bool wantse = sh[7] != 0.0;
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < len; i += stride) {
        const double t = (double)w[i];
        x[i] = (VT)(t1 * t + (double)x[i]);
        const VT wn = (VT)(t2 * t + (double)Vq[i] * sv);
        w[i] = wn;
        if (wantse) {
            const double d = (t3 * t) * (t3 * t);
            se[i] = (VT)(d + (double)se[i]);
        }
        s += (double)wn * (double)wn;
    }
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(VEC_BLOCK) void k_shard_s3w(const double *__restrict__ partials, int np,
                                                         double *__restrict__ wsq, const double *__restrict__ sums,
                                                         LsqrState *st, const int *__restrict__ live,
                                                         const void *__restrict__ x, int f32, double *__restrict__ log)
{
    __shared__ double red[VEC_BLOCK / WAVE];
    const double s = np > 0 ? strided_sum<VEC_BLOCK>(partials, np) : 0.0;
    const double tot = block_sum<VEC_BLOCK>(s, red);   // (k_reduce_partials' reduction, bit for bit)
    if (threadIdx.x != 0) return;
    wsq[0] = tot;
    if (*live == 0) return;
    const double x1 = x == nullptr ? 0.0 : (f32 ? (double)static_cast<const float *>(x)[0] : static_cast<const double *>(x)[0]);
    s3_step(st, (st->t3 * st->t3) * sums[1], x1, log);
}

}  // namespace lsqrhip

// stage ids (keep in sync with lsqr_amd/dist.py)
enum {
    ST_SUMSQ_B = 0,        // sums[0..2] = Blue's sums of b_p^2                       -> all-reduce sums[0..2]
    ST_INIT_BETA_ATU = 1,  // beta; T = A_p'(U_p / beta)                             -> exchange: slice q of T to rank q's R
    ST_INIT_V = 2,         // V_q = sum_r R[r]; sums[0] = |V_q ns|^2                  -> all-reduce sums[0..1]
    ST_INIT_W = 3,         // alpha; w_q = V_q / alpha; arnorm                        -> all-gather V
    ST_MODE1 = 4,          // U_p <- (-alpha)(U_p su) + A_p (V sv); sums[0] = |U_p ns|^2   -> all-reduce sums[0]
    ST_S1_ATU = 5,         // beta, anorm; T = A_p'(U_p su)                           -> exchange: slice q of T to rank q's R
    ST_VCOMBINE = 6,       // V_q <- (-beta)(V_q sv) + sum_r R[r]; sums[0] = |V_q ns|^2, sums[1] = |w_q|^2   -> all-reduce sums[0..1]
    ST_UPDATE = 7          // alpha, rotations; dknorm, tests; x_q, w_q, se_q         -> all-gather V
};

extern "C" int lsqrhip_shard_begin(lsqrhip_handle_t h, const double *d_b_local, int64_t m_global, int world, int rank,
                                   double damp, double atol, double btol, double conlim, int itnlim, int wantse,
                                   double *d_T, double *d_R, double *d_V, double *d_sums)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!d_T || !d_R || !d_V || !d_sums || (!d_b_local && h->m > 0)) return fail(LSQRHIP_ERR_ARG, "null shard buffer");
    if (world < 1 || rank < 0 || rank >= world) return fail(LSQRHIP_ERR_ARG, "bad world / rank");
    if (h->op || h->group) return fail(LSQRHIP_ERR_ARG, "the row-sharded stages need a matrix handle, not an operator or a sharded parent");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ShardCtx &c = h->shard;
    if (!c.engine_next) {   // not the engine's call: whatever an engine solve left behind (it may have failed before its
        c.own_in_T = false;   // lsqrhip_shard_end) must not steer this caller's stages -- they reduce `sums` themselves,
        c.gath = nullptr;     // copy their own slice of T to R and bring a 4-double `sums`
        c.vmax_msg = false;
        c.msg = 4;
    }
    c.engine_next = false;
    c.P = world;
    c.rank = rank;
    c.chunk = ((int64_t)h->n + world - 1) / world;
    c.my0 = std::min<int64_t>((int64_t)rank * c.chunk, h->n);
    c.mylen = std::min<int64_t>(c.chunk, (int64_t)h->n - c.my0);
    c.T = d_T; c.R = d_R; c.V = d_V; c.sums = d_sums;
    c.wantse = wantse;
    c.upar = 0;
    c.fuse_s2 = env_int("LSQRHIP_SHARD_FUSE_S2", 1) != 0;   // (once per solve, on the caller's thread: never from a stage)
    if (h->MXU != nullptr) HIPCHK(hipMemsetAsync(h->MXU, 0, sizeof(double) * 2 * MX_SET, s));   // (xmax_folded: both sets zero)
    if (!c.wsq) HIPCHK(hipMalloc((void **)&c.wsq, sizeof(double)));
    if (!c.live) HIPCHK(hipMalloc((void **)&c.live, sizeof(int)));
    RET(prepare_log(h, itnlim, c.want_log));
    RET(reset_csb_tickets(h));   // (solve_loop.h: a previous engine solve may have been abandoned between two phases of a product)
    LsqrState init;
    std::memset(&init, 0, sizeof(init));
    init.itnlim = itnlim;
    init.damped = damp > 0.0;
    init.wantse = wantse != 0;
    init.want_log = c.want_log != 0;   // the reference's log (src/lsqr.f90:813-837) from this rank's records
    init.log_cap = h->log_cap;
    init.m = (int)m_global;  // se finish uses the GLOBAL row count (src/lsqr.f90:857-861)
    init.n = h->n;
    init.damp = damp;
    init.atol = atol;
    init.btol = btol;
    init.ctol = conlim > 0.0 ? 1.0 / conlim : 0.0;
    init.cs2 = -1.0;
    init.su = init.sv = 1.0;
    init.ns_inv = h->nsc.inv;
    init.wp32 = h->f32 ? 1 : 0;
    init.c1.skip = init.c2.skip = init.c2p.skip = 1;
    *h->h_state = init;
    HIPCHK(hipMemcpyAsync(h->d_state, h->h_state, sizeof(LsqrState), hipMemcpyHostToDevice, s));
    // (a REAL32 handle: b, the exchange buffers T, R, V and the slices are float arrays -- half the bytes on the links)
    const size_t esz = h->f32 ? sizeof(float) : sizeof(double);
    const size_t m = (size_t)h->m, full = (size_t)(c.chunk * world);
    const size_t sl = std::min((size_t)std::max<int64_t>(c.chunk, 0), (size_t)h->n);
    if (m > 0) HIPCHK(hipMemcpyAsync(h->U, d_b_local, esz * m, hipMemcpyDeviceToDevice, s));
    if (full > 0) {
        HIPCHK(hipMemsetAsync(c.V, 0, esz * full, s));
        HIPCHK(hipMemsetAsync(c.T, 0, esz * full, s));
        HIPCHK(hipMemsetAsync(c.R, 0, esz * full, s));
    }
    if (sl > 0) {   // x_q, w_q, se_q live at the start of the handle's n-vectors
        HIPCHK(hipMemsetAsync(h->X, 0, esz * sl, s));
        HIPCHK(hipMemsetAsync(h->W, 0, esz * sl, s));
        if (wantse) HIPCHK(hipMemsetAsync(h->SE, 0, esz * sl, s));
    }
    HIPCHK(hipMemsetAsync(d_sums, 0, (c.vmax_msg ? SHARD_MSG : 4) * sizeof(double), s));
    HIPCHK(hipMemsetAsync(c.wsq, 0, sizeof(double), s));
    HIPCHK(hipMemsetAsync(c.live, 0, sizeof(int), s));
    c.active = true;
    return LSQRHIP_OK;
}

// this rank's slice of v and its x, w, se (element type: the handle's)
struct VecPtrs {
    void *X, *W, *SE;
    const void *Vq;
};
static VecPtrs shard_vec_ptrs(H *h)
{
    ShardCtx &c = h->shard;
    const size_t esz = h->f32 ? sizeof(float) : sizeof(double);
    return VecPtrs{h->X, h->W, h->SE, reinterpret_cast<const char *>(c.V) + esz * (size_t)c.my0};
}

// the kernels of a stage that touch the vectors, for binary64 and REAL32 handles alike
template <typename VT>
static void shard_stage_vec(H *h, int stage)
{
    hipStream_t s = h->stream;
    LsqrState *st = h->d_state;
    ShardCtx &c = h->shard;
    VT *Vq = reinterpret_cast<VT *>(c.V) + c.my0;
    const VT *R = reinterpret_cast<const VT *>(c.R);
    // (the C++ engine leaves the rank's own slice of T where it is: shard_engine.h ex_scatter)
    const VT *own = c.own_in_T ? reinterpret_cast<const VT *>(c.T) + c.my0 : nullptr;
    VT *X = reinterpret_cast<VT *>(h->X), *W = reinterpret_cast<VT *>(h->W), *SE = reinterpret_cast<VT *>(h->SE);
    const int gq = vec_grid(2 * std::max<int64_t>(c.mylen, 1));
    switch (stage) {
    case ST_SUMSQ_B:
        hipLaunchKernelGGL(k_sumsq3<VT>, dim3(h->vgrid_m), dim3(VEC_BLOCK), 0, s, (const VT *)reinterpret_cast<VT *>(h->U),
                           (int64_t)h->m, h->partials);
        break;
    case ST_INIT_V:
        hipLaunchKernelGGL(k_rs_combine<VT>, dim3(gq), dim3(VEC_BLOCK), 0, s, Vq, R, own, c.rank, c.P, c.chunk, c.mylen,
                           (const SpmvCoef *)&st->c2, (const int *)h->d_zero, h->partials, h->nsc,
                           c.vmax_msg ? h->partials + SPMV_MAX_GRID : (double *)nullptr);
        break;
    case ST_INIT_W:
        hipLaunchKernelGGL(k_init_w_slice<VT>, dim3(gq), dim3(VEC_BLOCK), 0, s, W, (const VT *)Vq, c.mylen,
                           (const LsqrState *)st, h->partials);
        break;
    case ST_VCOMBINE:
        hipLaunchKernelGGL(k_rs_combine<VT>, dim3(gq), dim3(VEC_BLOCK), 0, s, Vq, R, own, c.rank, c.P, c.chunk, c.mylen,
                           (const SpmvCoef *)&st->c2, (const int *)&st->stop, h->partials, h->nsc,
                           c.vmax_msg ? h->partials + SPMV_MAX_GRID : (double *)nullptr);
        break;
    case ST_UPDATE:
        hipLaunchKernelGGL(k_update_slice<VT>, dim3(gq), dim3(VEC_BLOCK), 0, s, X, W, (const VT *)Vq, SE, c.mylen,
                           (const LsqrState *)st, (const int *)c.live, h->partials);
        break;
    default:
        break;
    }
}

// Enqueue one stage on the handle's stream (asynchronous).  `phase` (shard_engine.h with LSQRHIP_SHARD_OVERLAP=1): the
// products of ST_MODE1 / ST_S1_ATU are launched one phase of their layout's plan at a time (csb.h "Column stripes /
// phases") -- the stage's scalar kernel goes with its first phase, the reduction of the partials with its last;
// -1: the whole stage.
static int shard_stage_phase(lsqrhip_handle_t h, int stage, int phase);
extern "C" int lsqrhip_shard_stage(lsqrhip_handle_t h, int stage) { return shard_stage_phase(h, stage, -1); }

static int shard_stage_phase(lsqrhip_handle_t h, int stage, int phase)
{
    if (!h || !h->shard.active) return fail(LSQRHIP_ERR_NOT_INIT, "lsqrhip_shard_begin was not called");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    LsqrState *st = h->d_state;
    ShardCtx &c = h->shard;
    double *T = c.T, *sums = c.sums;   // (T, V: float arrays in disguise for a REAL32 handle, like U, V, W, X)
    const int gq = vec_grid(2 * std::max<int64_t>(c.mylen, 1));
    auto vec = [&](int st_) {
        if (h->f32) shard_stage_vec<float>(h, st_);
        else shard_stage_vec<double>(h, st_);
    };
    switch (stage) {
    case ST_SUMSQ_B:
        vec(stage);
        hipLaunchKernelGGL(k_reduce_partials3, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials,
                           h->vgrid_m, sums);
        break;
    case ST_INIT_BETA_ATU:
        hipLaunchKernelGGL(k_s_init1<false>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, 0,
                           (const double *)sums, st, (NormSlot *)nullptr);
        launch_spmv(h, h->AT, h->U, T, &st->c2p, h->d_zero, nullptr, nullptr, true);
        break;
    case ST_INIT_V:
        vec(stage);
        hipLaunchKernelGGL(k_shard_sums, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials, gq,
                           (const double *)c.wsq, sums, (const SpmvCoef *)&st->c2,
                           c.vmax_msg ? (const double *)(h->partials + SPMV_MAX_GRID) : (const double *)nullptr);
        break;
    case ST_INIT_W:
        hipLaunchKernelGGL((k_s_init2<false>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, 0,
                           (const double *)sums, st);
        vec(stage);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials, gq, c.wsq);
        break;
    case ST_MODE1: {
        // (the C++ engine: the piece maxima of v came with the norms, P * SHARD_NMAX of them in xmax_part)
        SpmvArgs a;
        a.c = &h->A; a.x = c.V; a.y = h->U; a.coef = &st->c1; a.stop = &st->stop; a.pout = h->partials; a.stream = s;
        a.unit_x = true;
        a.phase = phase;
        if (c.vmax_msg) {
            a.xmax_in = h->xmax_part;
            a.nxmax_in = c.P * SHARD_NMAX;
        }
        if (c.gath != nullptr && xmax_folded(h)) {   // (the engine: u's piece maxima for mode 2, solve_loop.h xmax_folded)
            if (phase <= 0) c.upar ^= 1;
            a.ymax_out = h->MXU + c.upar * MX_SET;
        }
        launch_spmv_args(h, a);
        if (phase < 0 || phase >= std::max(h->A.csb ? h->A.phases : 1, 1) - 1)
            hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials,
                               h->A.out_grid, sums);
        break;
    }
    case ST_S1_ATU: {
        if (phase <= 0) {
            if (c.gath != nullptr)   // (the C++ engine: the rank-ordered sum of the gathered norms rides along)
                hipLaunchKernelGGL(k_shard_s1g, dim3(1), dim3(64), 0, s, (const double *)(c.P > 1 ? c.gath : sums), c.P,
                                   c.msg, st);
            else
                hipLaunchKernelGGL((k_s1<false>), dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, 0,
                                   (const double *)sums, st);
        }
        SpmvArgs a;
        a.c = &h->AT; a.x = h->U; a.y = T; a.coef = &st->c2p; a.stop = &st->stop; a.pout = h->partials; a.stream = s;
        a.unit_x = true;
        a.phase = phase;
        if (c.gath != nullptr && xmax_folded(h)) {
            a.xmax_in = h->MXU + c.upar * MX_SET;
            a.nxmax_in = csb_npieces(h->m);
            a.xmax_clr = h->MXU + (c.upar ^ 1) * MX_SET;   // (what the next mode 1 raises)
        }
        launch_spmv_args(h, a);
        break;
    }
    case ST_VCOMBINE:
        vec(stage);
        hipLaunchKernelGGL(k_shard_sums, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials, gq,
                           (const double *)c.wsq, sums, (const SpmvCoef *)&st->c2,
                           c.vmax_msg ? (const double *)(h->partials + SPMV_MAX_GRID) : (const double *)nullptr);
        break;
    case ST_UPDATE:
        if (c.gath != nullptr && c.fuse_s2) {   // step 2 inside the update's launch
            const double *src = c.P > 1 ? c.gath : sums;
            double *vmx = c.vmax_msg ? h->xmax_part : (double *)nullptr;
            const VecPtrs vp = shard_vec_ptrs(h);
            if (h->f32)
                hipLaunchKernelGGL(k_update_slice_g<float>, dim3(gq), dim3(VEC_BLOCK), 0, s, (float *)vp.X, (float *)vp.W,
                                   (const float *)vp.Vq, (float *)vp.SE, c.mylen, st, c.live, h->partials, src, c.P, c.msg,
                                   sums, vmx);
            else
                hipLaunchKernelGGL(k_update_slice_g<double>, dim3(gq), dim3(VEC_BLOCK), 0, s, (double *)vp.X, (double *)vp.W,
                                   (const double *)vp.Vq, (double *)vp.SE, c.mylen, st, c.live, h->partials, src, c.P, c.msg,
                                   sums, vmx);
        } else {
            if (c.gath != nullptr)
                hipLaunchKernelGGL(k_shard_s2g, dim3(1), dim3(64), 0, s, (const double *)(c.P > 1 ? c.gath : sums), c.P,
                                   c.msg, sums, st, c.live, c.vmax_msg ? h->xmax_part : (double *)nullptr);
            else
                hipLaunchKernelGGL(k_shard_s2, dim3(1), dim3(1), 0, s, (const double *)sums, st, c.live);
            vec(stage);
        }
        // step 3 AFTER the update, as in the reference (src/lsqr.f90:729-745, then :751-837): the x(1) of the
        // iteration log is the updated one; both are gated by `live`, which step 2 set for this iteration
        if (c.gath != nullptr) {
            hipLaunchKernelGGL(k_shard_s3w, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials, gq, c.wsq,
                               (const double *)sums, st, (const int *)c.live, (const void *)h->X, h->f32 ? 1 : 0, h->d_log);
        } else {
            hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials, gq, c.wsq);
            hipLaunchKernelGGL(k_shard_s3, dim3(1), dim3(1), 0, s, (const double *)sums, st, (const int *)c.live,
                               (const void *)h->X, h->f32 ? 1 : 0, h->d_log);
        }
        break;
    default:
        return fail(LSQRHIP_ERR_ARG, "unknown shard stage");
    }
    HIPCHK(hipGetLastError());
    return LSQRHIP_OK;
}

// d_out[0..chunk) = sum over r < nchunks of d_in[r*chunk .. (r+1)*chunk), in rank order; asynchronous
// on the handle's stream.
extern "C" int lsqrhip_sum_chunks(lsqrhip_handle_t h, const double *d_in, int nchunks, int64_t chunk, double *d_out)
{
    if (!h || !d_in || !d_out || nchunks < 1 || chunk < 0) return fail(LSQRHIP_ERR_ARG, "bad sum_chunks arguments");
    if (h->f32) return fail(LSQRHIP_ERR_ARG, "lsqrhip_sum_chunks sums binary64 vectors (not for a REAL32 handle)");
    HIPCHK(hipSetDevice(h->device));
    if (chunk > 0)
        hipLaunchKernelGGL(k_sum_chunks, dim3(vec_grid(2 * chunk)), dim3(VEC_BLOCK), 0, h->stream, d_out, d_in, nchunks,
                           chunk);
    HIPCHK(hipGetLastError());
    return LSQRHIP_OK;
}

// Wait for the stream and report the loop state: out[0] = stop, out[1] = itn, out[2] = istop.
extern "C" int lsqrhip_shard_poll(lsqrhip_handle_t h, int *out)
{
    if (!h || !out) return fail(LSQRHIP_ERR_ARG, "null handle or buffer");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(LsqrState), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    out[0] = h->h_state->stop;
    out[1] = h->h_state->itn;
    out[2] = h->h_state->istop;
    return LSQRHIP_OK;
}

// Finish: se, istop 2 -> 3; this rank's slices of x (and se) are written at their place
// [rank * chunk, ...) of the caller's P * chunk buffers (the caller all-gathers them); scalar outputs.
extern "C" int lsqrhip_shard_end(lsqrhip_handle_t h, double *d_x, double *d_se, int *istop, int *itn, double *anorm,
                                 double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    if (!h || !h->shard.active) return fail(LSQRHIP_ERR_NOT_INIT, "lsqrhip_shard_begin was not called");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    ShardCtx &c = h->shard;
    const size_t len = (size_t)std::max<int64_t>(c.mylen, 0);
    const size_t esz = h->f32 ? sizeof(float) : sizeof(double);
    if (c.wantse && len > 0) {
        if (h->f32)
            hipLaunchKernelGGL(k_se_finish<float>, dim3(vec_grid(2 * (int64_t)len)), dim3(VEC_BLOCK), 0, s, (float *)h->SE,
                               (int64_t)len, (const LsqrState *)h->d_state);
        else
            hipLaunchKernelGGL(k_se_finish<double>, dim3(vec_grid(2 * (int64_t)len)), dim3(VEC_BLOCK), 0, s, h->SE,
                               (int64_t)len, (const LsqrState *)h->d_state);
    }
    if (d_x && len > 0)
        HIPCHK(hipMemcpyAsync((char *)d_x + esz * (size_t)c.my0, h->X, esz * len, hipMemcpyDeviceToDevice, s));
    if (d_se && c.wantse && len > 0)
        HIPCHK(hipMemcpyAsync((char *)d_se + esz * (size_t)c.my0, h->SE, esz * len, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(LsqrState), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const LsqrState &r = *h->h_state;
    if (c.want_log && r.itn > 0) {   // as finish_solve does for the one-GPU path
        h->log_count = std::min(r.log_count, h->log_cap);
        h->h_log.resize((size_t)h->log_count * LOG_STRIDE);
        HIPCHK(hipMemcpy(h->h_log.data(), h->d_log, sizeof(double) * h->h_log.size(), hipMemcpyDeviceToHost));
    }
    int is = r.istop;
    if (r.damped && is == 2) is = 3;
    if (istop) *istop = is;
    if (itn) *itn = r.itn;
    if (anorm) *anorm = r.anorm;
    if (acond) *acond = r.acond;
    if (rnorm) *rnorm = r.rnorm;
    if (arnorm) *arnorm = r.arnorm;
    if (xnorm) *xnorm = r.xnorm;
    c.active = false;
    c.own_in_T = false;
    c.gath = nullptr;
    c.vmax_msg = false;   // (the next caller of lsqrhip_shard_begin may bring a 4-double `sums`: lsqr_amd/dist.py)
    c.msg = 4;
    return LSQRHIP_OK;
}
