// shard_api.h -- row-block sharded LSQR: the per-rank stages (included by lsqrhip.hip).
//
// One process per GPU holds A_p = rows [row0, row0 + m_p) of A (its own handle).  u and b are
// sharded with the rows; v, w, x (n-vectors) are replicated.  Per iteration the ONLY exchanges
// are:
//     sum_p |u_p|^2        one double, all-reduce   (beta)
//     sum_p A_p' u_p       n doubles,  all-reduce   (the n-vector after the A'-apply)
// Everything else is local, and because every rank then holds identical v and identical
// all-reduced sums, the replicated scalar recurrences stay bit-identical across ranks.
// The collectives themselves are issued by the host (torch.distributed = RCCL over xGMI,
// lsqr_amd/dist.py) on buffers it owns (T, sums); this file only launches local kernels on
// the handle's stream, asynchronously.
#pragma once

namespace lsqrhip {

// V <- cy*(V*sy) + T ; partial sums of V^2          (the "combine" after the all-reduce)
__global__ __launch_bounds__(VEC_BLOCK) void k_vcombine(double *__restrict__ V, const double *__restrict__ T,
                                                        int64_t n, const SpmvCoef *__restrict__ coef,
                                                        const int *__restrict__ stop,
                                                        double *__restrict__ partials, NScale nsc)
{
    if (*stop != 0 || coef->skip != 0) return;
    const double sy = coef->sy, cy = coef->cy;
    __shared__ double red[VEC_BLOCK / WAVE];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n; i += stride) {
        const double v = cy * (V[i] * sy) + T[i];
        V[i] = v;
        const double vs = v * nsc.s;
        s += vs * vs;
    }
    const double tot = block_sum<VEC_BLOCK>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// out[i] = in[0*chunk + i] + in[1*chunk + i] + ... in rank order (the local sum of a direct
// reduce-scatter: every element is summed once, by its owner, in a fixed order)
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

}  // namespace lsqrhip

// stage ids (keep in sync with lsqr_amd/dist.py)
enum { ST_SUMSQ_B = 0, ST_INIT_BETA_ATU = 1, ST_INIT_V = 2, ST_MODE1 = 3, ST_S1_ATU = 4, ST_VCOMBINE_UPDATE = 5 };

extern "C" int lsqrhip_shard_begin(lsqrhip_handle_t h, const double *d_b_local, int64_t m_global, double damp,
                                   double atol, double btol, double conlim, int itnlim, int wantse, double *d_T,
                                   double *d_sums)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (!d_T || !d_sums || (!d_b_local && h->m > 0)) return fail(LSQRHIP_ERR_ARG, "null shard buffer");
    if (h->op) return fail(LSQRHIP_ERR_ARG, "the row-sharded solve needs a matrix handle, not an operator");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    LsqrState init;
    std::memset(&init, 0, sizeof(init));
    init.itnlim = itnlim;
    init.damped = damp > 0.0;
    init.wantse = wantse != 0;
    init.m = (int)m_global;  // se finish uses the GLOBAL row count (src/lsqr.f90:857-861)
    init.n = h->n;
    init.damp = damp;
    init.atol = atol;
    init.btol = btol;
    init.ctol = conlim > 0.0 ? 1.0 / conlim : 0.0;
    init.cs2 = -1.0;
    init.su = init.sv = 1.0;
    init.ns_inv = h->nsc.inv;
    init.c1.skip = init.c2.skip = init.c2p.skip = 1;
    *h->h_state = init;
    HIPCHK(hipMemcpyAsync(h->d_state, h->h_state, sizeof(LsqrState), hipMemcpyHostToDevice, s));
    const size_t n = (size_t)h->n, m = (size_t)h->m;
    if (m > 0) HIPCHK(hipMemcpyAsync(h->U, d_b_local, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    if (n > 0) {
        HIPCHK(hipMemsetAsync(h->V, 0, sizeof(double) * n, s));
        HIPCHK(hipMemsetAsync(h->X, 0, sizeof(double) * n, s));
        HIPCHK(hipMemsetAsync(h->W, 0, sizeof(double) * n, s));
        HIPCHK(hipMemsetAsync(d_T, 0, sizeof(double) * n, s));
        if (wantse) HIPCHK(hipMemsetAsync(h->SE, 0, sizeof(double) * n, s));
    }
    HIPCHK(hipMemsetAsync(d_sums, 0, 4 * sizeof(double), s));
    h->shard_T = d_T;
    h->shard_sums = d_sums;
    h->shard_wantse = wantse;
    return LSQRHIP_OK;
}

// Enqueue one stage on the handle's stream (asynchronous).
extern "C" int lsqrhip_shard_stage(lsqrhip_handle_t h, int stage)
{
    if (!h || !h->shard_T) return fail(LSQRHIP_ERR_NOT_INIT, "lsqrhip_shard_begin was not called");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    LsqrState *st = h->d_state;
    double *T = h->shard_T, *sums = h->shard_sums;
    const int64_t n = h->n, m = h->m;
    switch (stage) {
    case ST_SUMSQ_B:  // sums[0..2] = Blue's small / mid / big sums of b_p^2 (range-safe, additive over ranks)
        hipLaunchKernelGGL(k_sumsq3, dim3(h->vgrid_m), dim3(VEC_BLOCK), 0, s, (const double *)h->U, m, h->partials);
        hipLaunchKernelGGL(k_reduce_partials3, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials,
                           h->vgrid_m, sums);
        break;
    case ST_INIT_BETA_ATU:  // beta from the all-reduced sums[0..2]; T_p = A_p'(U_p/beta)
        hipLaunchKernelGGL(k_s_init1<false>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, 0,
                           (const double *)sums, st, (NormSlot *)nullptr);
        launch_spmv(h, h->AT, h->U, T, &st->c2p, h->d_zero, nullptr, nullptr, true);
        break;
    case ST_INIT_V:  // V = sum_p T_p (all-reduced); alpha, v, w
        hipLaunchKernelGGL(k_vcombine, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->V, (const double *)T, n,
                           (const SpmvCoef *)&st->c2, (const int *)h->d_zero, h->partials, h->nsc);
        hipLaunchKernelGGL(k_s_init2<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                           (const double *)nullptr, st);
        hipLaunchKernelGGL(k_copy_scale, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->W, (const double *)h->V, n,
                           (const LsqrState *)st);
        break;
    case ST_MODE1:  // U_p <- (-alpha)(U_p su) + A_p (V sv); sums[0] = |U_p|^2
        launch_spmv(h, h->A, h->V, h->U, &st->c1, &st->stop, nullptr, nullptr, true);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(VEC_BLOCK), 0, s, (const double *)h->partials,
                           h->A.out_grid, sums);
        break;
    case ST_S1_ATU:  // beta, anorm from the all-reduced sums[0]; T_p = A_p'(U_p su)
        hipLaunchKernelGGL(k_s1<false>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, 0,
                           (const double *)sums, st);
        launch_spmv(h, h->AT, h->U, T, &st->c2p, &st->stop, nullptr, nullptr, true);
        break;
    case ST_VCOMBINE_UPDATE:  // V <- (-beta)(V sv) + sum_p T_p; alpha; rotations; x, w; tests
        hipLaunchKernelGGL(k_vcombine, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->V, (const double *)T, n,
                           (const SpmvCoef *)&st->c2, (const int *)&st->stop, h->partials, h->nsc);
        hipLaunchKernelGGL(k_s2<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                           (const double *)nullptr, st);
        hipLaunchKernelGGL(k_update, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->X, h->W, (const double *)h->V,
                           h->SE, n, (const LsqrState *)st, h->partials);
        hipLaunchKernelGGL(k_s3<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                           (const double *)nullptr, st, (const double *)h->X, h->d_log);
        break;
    default:
        return fail(LSQRHIP_ERR_ARG, "unknown shard stage");
    }
    HIPCHK(hipGetLastError());
    return LSQRHIP_OK;
}

// d_out[0..chunk) = sum over r < nchunks of d_in[r*chunk .. (r+1)*chunk), in rank order; asynchronous
// on the handle's stream.  The local step of the direct reduce-scatter in lsqr_amd/dist.py.
extern "C" int lsqrhip_sum_chunks(lsqrhip_handle_t h, const double *d_in, int nchunks, int64_t chunk, double *d_out)
{
    if (!h || !d_in || !d_out || nchunks < 1 || chunk < 0) return fail(LSQRHIP_ERR_ARG, "bad sum_chunks arguments");
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

// Finish: se, istop 2 -> 3, copy x (and se) to device buffers of the caller, scalar outputs.
extern "C" int lsqrhip_shard_end(lsqrhip_handle_t h, double *d_x, double *d_se, int *istop, int *itn, double *anorm,
                                 double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    if (!h || !h->shard_T) return fail(LSQRHIP_ERR_NOT_INIT, "lsqrhip_shard_begin was not called");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t n = (size_t)h->n;
    if (h->shard_wantse && n > 0)
        hipLaunchKernelGGL(k_se_finish, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->SE, (int64_t)n,
                           (const LsqrState *)h->d_state);
    if (d_x && n > 0) HIPCHK(hipMemcpyAsync(d_x, h->X, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    if (d_se && h->shard_wantse && n > 0) HIPCHK(hipMemcpyAsync(d_se, h->SE, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(h->h_state, h->d_state, sizeof(LsqrState), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const LsqrState &r = *h->h_state;
    int is = r.istop;
    if (r.damped && is == 2) is = 3;
    if (istop) *istop = is;
    if (itn) *itn = r.itn;
    if (anorm) *anorm = r.anorm;
    if (acond) *acond = r.acond;
    if (rnorm) *rnorm = r.rnorm;
    if (arnorm) *arnorm = r.arnorm;
    if (xnorm) *xnorm = r.xnorm;
    h->shard_T = nullptr;
    h->shard_sums = nullptr;
    return LSQRHIP_OK;
}
