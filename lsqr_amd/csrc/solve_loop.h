// solve_loop.h -- the device-resident LSQR loop (included by lsqrhip.hip).
//
// Three schedules over the same arithmetic (results are bit-identical; tests compare them):
//
//  sequential ("pipeline" = 0)      K1 -> S1 -> K2 -> S2 -> K4 -> S3            6 launches / iteration
//
//  riders     ("pipeline" = 1)                                                   3 launches / iteration
//      K1(i+1) (+) S12(i)  ->  K4(i)  ->  K2(i+1) (+) S3(i)  ->  K1(i+2) (+) S12(i+1)  -> ...
//
//      K1 = mode-1 SpMV, K2 = mode-2 SpMV, K4 = x/w update, S* = the scalar steps (scalar.h).
//      At config 2 each scalar kernel is ~4.7 us (a chain of fp64 divides and square roots
//      behind a kernel boundary) during which HBM idles: 14 us of a 58 us iteration.  Here:
//        * K2 derives beta from K1's partials and K1 derives alpha from K2's partials in a
//          short prologue (spmv.h, lazy coefficients), so nothing sits between the SpMVs;
//        * the authoritative scalar machine (anorm, rotations, norm estimates, stopping
//          tests) runs as a RIDER: one extra workgroup of the next SpMV launch.  A rider's
//          inputs are complete when its host kernel starts and its outputs are first read by
//          a later kernel, so stream order is the only synchronisation -- no flags, no
//          fences, one stream, one queue.
//      K1/K2 launched past the stopping iteration are speculative: they only touch U, V and
//      partials; K4 and the riders check `stop` (written by the S3 rider two launches
//      earlier), so x, w, se and the scalar state are exactly those of the stopping iteration.
//      Partials and hand-over slots are double-buffered by iteration parity; batches hold an
//      even number of iterations so a captured graph replays with the right parity.  Each
//      batch ends with the last iteration's S12 -> K4 -> S3 as plain kernels so that the host
//      polls a settled state.
//
//  fused update ("pipeline" = 2, default; needs a non-panelled A)               2 launches / iteration
//      K1(i+1) (+) S12(i) (+) K4(i)  ->  K2(i+1) (+) S3(i)  ->  ...
//
//      K4(i) needs t1, t2, t3 of iteration i, which S12(i) computes in the SAME launch.  Every
//      workgroup therefore evaluates the rotation itself (scalar.h rot_step: a dozen flops)
//      from inputs that are complete before the launch: alpha (its own lazy norm), beta (the
//      slot mode 2 published) and rhobar / phibar of iteration i-1, which the state keeps by
//      iteration parity so that the rider's writes for iteration i land in the other slot.
//      Same function, same inputs => the same bits as the scalar machine.  Each workgroup then
//      runs whole blocks of k_update's own decomposition (vec.h update_block), so the update
//      partials -- and with them dnorm, acond and the stopping tests -- are bit-identical to
//      the separate kernel's.  Removes a launch boundary and lets the update's streaming
//      traffic overlap the latency-bound SpMV phases.
//
//      (A two-stream variant -- scalar machine and K4 on a side stream beside the next
//      SpMV -- was measured first: 73 us/iteration device time against 60 sequential.
//      Cross-queue dependencies inside a hipGraph cost more than the bubbles they hide.)
#pragma once
#include <atomic>

// ---------------------------------------------------------------------------
// kernel launch helpers
// ---------------------------------------------------------------------------
struct SpmvArgs {
    const Csr *c = nullptr;
    const double *x = nullptr;
    double *y = nullptr;
    const SpmvCoef *coef = nullptr;  // explicit coefficients ...
    const double *pin = nullptr;     // ... or lazy: partials to derive the norm from
    int npin = 0;
    const NormSlot *slot_in = nullptr;
    NormSlot *slot_out = nullptr;
    int skip_if_zero = 0;
    const int *stop = nullptr;
    double *pout = nullptr;
    Rider rider{};                   // scalar work carried by one extra workgroup (kind 0 = none)
    UpdArgs upd{};                   // x/w update of the previous iteration carried by this launch (on = 0: none)
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;  // kernel begin/end timestamps (hipExtLaunchKernelGGL)
    NScale nsc{1.0, 1.0};            // power-of-two scale inside the sums of squares (filled in by launch_spmv_args)
    bool f32 = false;                // x, y, the values and the update's vectors are float arrays (REAL32 handle)
    bool unit_x = false;             // |x * sx| <= 1 is known (solver-internal vectors)
    const double *xmax_in = nullptr; // csb.h: piece maxima of |x| the caller already holds (the sharded engine: each rank's
    int nxmax_in = 0;                // share travels with the norms) -- the k_csb_xmax pass over x is then left out
    double *ymax_out = nullptr;      // csb.h: where this product raises the piece maxima of the y it writes (csb_npieces(rows)
                                     // words, zero on entry), for the product that gathers from it next -- or null
    double *xmax_clr = nullptr;      // csb.h: with xmax_in, the other set of x's piece maxima: zeroed on the way (CsbX.clr)
    int phase = -1, nphases = 1;     // csb.h with column stripes: launch only the sweeps of this phase (-1: the whole product)
};

template <typename OffT, bool PANEL, bool C16, bool V8, bool UPD, bool XL = false, typename VT = double>
static void launch_spmv_C(const SpmvArgs &a, double *y_, hipEvent_t e0, hipEvent_t e1)
{
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(y_);
    const Csr &c = *a.c;
    const dim3 grid(c.grid + (a.rider.kind != 0 ? 1 : 0));
    const void *colv = C16 ? (const void *)c.col16 : (const void *)c.col;
    const void *valv = V8 ? (const void *)c.val8 : (const void *)c.val;
    const XlArgs xa{c.rows, c.pw, c.cols, c.skew, c.rel16};
    if (e0 == nullptr && e1 == nullptr)  // plain launch (the only form used under stream capture)
        hipLaunchKernelGGL((k_spmv_fused<OffT, PANEL, C16, V8, UPD, XL, VT>), grid, dim3(SPMV_BLOCK), 0, a.stream,
                           (const OffT *)c.rowptr, colv, (const int *)c.cbase, valv, (const double *)c.dict,
                           (const RowBlock *)c.blk, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in,
                           a.slot_out, a.skip_if_zero, a.rider, a.upd, xa, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_fused<OffT, PANEL, C16, V8, UPD, XL, VT>), grid, dim3(SPMV_BLOCK), 0, a.stream, e0, e1, 0,
                              (const OffT *)c.rowptr, colv, (const int *)c.cbase, valv, (const double *)c.dict,
                              (const RowBlock *)c.blk, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin,
                              a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd, xa, a.nsc);
}

// row windows of a REAL32 handle (never panelled)
template <typename OffT>
static void launch_spmv_F(const SpmvArgs &a, double *y, hipEvent_t e0, hipEvent_t e1)
{
    const bool v8 = a.c->val8 != nullptr, c16 = a.c->col16 != nullptr;
    const int key = (c16 ? 4 : 0) | (v8 ? 2 : 0) | (a.upd.on ? 1 : 0);
    switch (key) {
    case 7: launch_spmv_C<OffT, false, true, true, true, false, float>(a, y, e0, e1); break;
    case 6: launch_spmv_C<OffT, false, true, true, false, false, float>(a, y, e0, e1); break;
    case 5: launch_spmv_C<OffT, false, true, false, true, false, float>(a, y, e0, e1); break;
    case 4: launch_spmv_C<OffT, false, true, false, false, false, float>(a, y, e0, e1); break;
    case 3: launch_spmv_C<OffT, false, false, true, true, false, float>(a, y, e0, e1); break;
    case 2: launch_spmv_C<OffT, false, false, true, false, false, float>(a, y, e0, e1); break;
    case 1: launch_spmv_C<OffT, false, false, false, true, false, float>(a, y, e0, e1); break;
    default: launch_spmv_C<OffT, false, false, false, false, false, float>(a, y, e0, e1); break;
    }
}

template <typename OffT, bool PANEL>
static void launch_spmv_T(const SpmvArgs &a, double *y, hipEvent_t e0, hipEvent_t e1)
{
    const bool v8 = a.c->val8 != nullptr;
    if (!PANEL && a.upd.on) {  // mode 1 carrying the x/w update (never a panelled product)
        if (a.c->col16 != nullptr) {
            if (v8) launch_spmv_C<OffT, false, true, true, true>(a, y, e0, e1);
            else launch_spmv_C<OffT, false, true, false, true>(a, y, e0, e1);
        } else {
            if (v8) launch_spmv_C<OffT, false, false, true, true>(a, y, e0, e1);
            else launch_spmv_C<OffT, false, false, false, true>(a, y, e0, e1);
        }
        return;
    }
    if (PANEL && a.c->xlds) {  // LDS-resident panels (spmv.h XL)
        if (v8) launch_spmv_C<OffT, PANEL, false, true, false, PANEL>(a, y, e0, e1);
        else launch_spmv_C<OffT, PANEL, false, false, false, PANEL>(a, y, e0, e1);
        return;
    }
    if (!PANEL && a.c->col16 != nullptr) {
        if (v8) launch_spmv_C<OffT, false, true, true, false>(a, y, e0, e1);
        else launch_spmv_C<OffT, false, true, false, false>(a, y, e0, e1);
    } else {
        if (v8) launch_spmv_C<OffT, PANEL, false, true, false>(a, y, e0, e1);
        else launch_spmv_C<OffT, PANEL, false, false, false>(a, y, e0, e1);
    }
}

template <bool C16, bool V8, bool UPD, typename VT, bool NT>
static void launch_sell_N(const SpmvArgs &a)
{
    const Csr &c = *a.c;
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(a.y);
    const dim3 grid(c.grid + (a.rider.kind != 0 ? 1 : 0));
    if (a.e0 == nullptr && a.e1 == nullptr)
        hipLaunchKernelGGL((k_spmv_sell<C16, V8, UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, (const unsigned *)c.soff,
                           (const void *)c.scol, (const int *)c.cbaseS, (const void *)c.sval, (const double *)c.dict,
                           (const unsigned char *)c.rlen, c.rows, c.nslices, c.nblk, x, y, a.coef, a.stop, a.pout,
                           a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_sell<C16, V8, UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, a.e0, a.e1, 0,
                              (const unsigned *)c.soff, (const void *)c.scol, (const int *)c.cbaseS,
                              (const void *)c.sval, (const double *)c.dict, (const unsigned char *)c.rlen, c.rows,
                              c.nslices, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in,
                              a.slot_out, a.skip_if_zero, a.rider, a.upd, a.nsc);
}

// (the matrix stream non-temporal or not: Csr.nt, common.h ld_stream)
template <bool C16, bool V8, bool UPD, typename VT = double>
static void launch_sell_C(const SpmvArgs &a)
{
    if (a.c->nt) launch_sell_N<C16, V8, UPD, VT, true>(a);
    else launch_sell_N<C16, V8, UPD, VT, false>(a);
}

template <bool UPD, typename VT, bool NT>
static void launch_sellp_N(const SpmvArgs &a)
{
    const Csr &c = *a.c;
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(a.y);
    const dim3 grid(c.grid + (a.rider.kind != 0 ? 1 : 0));
    if (a.e0 == nullptr && a.e1 == nullptr)
        hipLaunchKernelGGL((k_spmv_sellp<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, (const unsigned *)c.soff,
                           (const uint4 *)c.srec, (const int *)c.cbaseS, (const double *)c.dict, c.rows, c.nslices,
                           c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out,
                           a.skip_if_zero, a.rider, a.upd, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_sellp<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, a.e0, a.e1, 0,
                              (const unsigned *)c.soff, (const uint4 *)c.srec, (const int *)c.cbaseS,
                              (const double *)c.dict, c.rows, c.nslices, c.nblk, x, y, a.coef, a.stop, a.pout,
                              a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd, a.nsc);
}

template <bool UPD, typename VT = double>
static void launch_sellp(const SpmvArgs &a)
{
    if (a.c->nt) launch_sellp_N<UPD, VT, true>(a);
    else launch_sellp_N<UPD, VT, false>(a);
}

template <bool UPD, typename VT, bool NT, int U>
static void launch_pat2_U(const SpmvArgs &a, const dim3 grid)
{
    const Csr &c = *a.c;
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(a.y);
    if (a.e0 == nullptr && a.e1 == nullptr)
        hipLaunchKernelGGL((k_spmv_pat2<UPD, VT, NT, U>), grid, dim3(SELL_BLOCK), 0, a.stream,
                           (const unsigned short *)c.pid, (const PatEnt *)c.pent, c.pat_stride, c.rows,
                           c.nslices, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out,
                           a.skip_if_zero, a.rider, a.upd, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_pat2<UPD, VT, NT, U>), grid, dim3(SELL_BLOCK), 0, a.stream, a.e0, a.e1, 0,
                              (const unsigned short *)c.pid, (const PatEnt *)c.pent, c.pat_stride, c.rows,
                              c.nslices, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out,
                              a.skip_if_zero, a.rider, a.upd, a.nsc);
}

template <bool UPD, typename VT, bool NT, int U>
static void launch_patp_U(const SpmvArgs &a, const dim3 grid)
{
    const Csr &c = *a.c;
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(a.y);
    if (a.e0 == nullptr && a.e1 == nullptr)
        hipLaunchKernelGGL((k_spmv_patp<UPD, VT, NT, U>), grid, dim3(SELL_BLOCK), 0, a.stream, (const unsigned char *)c.pid,
                           (const unsigned *)c.pdesc, (const int *)c.pdelta, (const double *)c.pval, c.npat_e, c.rows,
                           c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero,
                           a.rider, a.upd, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_patp<UPD, VT, NT, U>), grid, dim3(SELL_BLOCK), 0, a.stream, a.e0, a.e1, 0,
                              (const unsigned char *)c.pid, (const unsigned *)c.pdesc, (const int *)c.pdelta,
                              (const double *)c.pval, c.npat_e, c.rows, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin,
                              a.npin, a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd, a.nsc);
}

template <bool UPD, typename VT, bool NT>
static void launch_pat_N(const SpmvArgs &a)
{
    const Csr &c = *a.c;
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(a.y);
    const dim3 grid(c.grid + (a.rider.kind != 0 ? 1 : 0));
    if (c.pat_wide && c.pat_pair) {   // ... in paired rows
        if (a.e0 == nullptr && a.e1 == nullptr)
            hipLaunchKernelGGL((k_spmv_pat2p<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream,
                               (const unsigned short *)c.pid, (const PatEnt *)c.pent, c.pat_stride, c.rows, c.nblk, x, y,
                               a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd,
                               a.nsc);
        else
            hipExtLaunchKernelGGL((k_spmv_pat2p<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, a.e0, a.e1, 0,
                                  (const unsigned short *)c.pid, (const PatEnt *)c.pent, c.pat_stride, c.rows, c.nblk, x, y,
                                  a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, a.rider,
                                  a.upd, a.nsc);
        return;
    }
    if (c.pat_wide) {   // two-byte pattern numbers, the table in global memory (pat.h "wide")
        switch (c.pat_u) {
        case 1: launch_pat2_U<UPD, VT, NT, 1>(a, grid); break;
        default: launch_pat2_U<UPD, VT, NT, 2>(a, grid); break;
        }
        return;
    }
    if (c.pat_pair) {   // lane L owns rows 2L, 2L + 1 (pat.h "paired rows")
        // (two groups per trip: 128 registers, 46k against 49k it/s at config 2)
        launch_patp_U<UPD, VT, NT, 1>(a, grid);
        return;
    }
    if (a.e0 == nullptr && a.e1 == nullptr)
        hipLaunchKernelGGL((k_spmv_pat<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, (const unsigned char *)c.pid,
                           (const unsigned *)c.pdesc, (const int *)c.pdelta, (const double *)c.pval, c.npat_e, c.rows,
                           c.nslices, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out,
                           a.skip_if_zero, a.rider, a.upd, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_pat<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, a.e0, a.e1, 0,
                              (const unsigned char *)c.pid, (const unsigned *)c.pdesc, (const int *)c.pdelta,
                              (const double *)c.pval, c.npat_e, c.rows, c.nslices, c.nblk, x, y, a.coef, a.stop,
                              a.pout, a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd, a.nsc);
}

template <bool UPD, typename VT = double>
static void launch_pat(const SpmvArgs &a)
{
    if (a.c->nt) launch_pat_N<UPD, VT, true>(a);
    else launch_pat_N<UPD, VT, false>(a);
}

template <bool UPD, typename VT, bool NT>
static void launch_spat_N(const SpmvArgs &a)
{
    const Csr &c = *a.c;
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(a.y);
    const dim3 grid(c.grid + (a.rider.kind != 0 ? 1 : 0));
    if (a.e0 == nullptr && a.e1 == nullptr)
        hipLaunchKernelGGL((k_spmv_spat<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, (const unsigned char *)c.pid,
                           (const unsigned *)c.pdesc, (const int *)c.pdelta, c.npat_e, (const unsigned *)c.soff,
                           (const VT *)c.sval, c.rows, c.nslices, c.nblk, x, y, a.coef, a.stop, a.pout, a.pin, a.npin,
                           a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_spat<UPD, VT, NT>), grid, dim3(SELL_BLOCK), 0, a.stream, a.e0, a.e1, 0,
                              (const unsigned char *)c.pid, (const unsigned *)c.pdesc, (const int *)c.pdelta, c.npat_e,
                              (const unsigned *)c.soff, (const VT *)c.sval, c.rows, c.nslices, c.nblk, x, y, a.coef,
                              a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, a.rider, a.upd,
                              a.nsc);
}

template <bool UPD, typename VT = double>
static void launch_spat(const SpmvArgs &a)
{
    if (a.c->nt) launch_spat_N<UPD, VT, true>(a);
    else launch_spat_N<UPD, VT, false>(a);
}

template <typename OffT, bool V8, bool C16>
static void launch_xl_C(const SpmvArgs &a, double *z)
{
    const Csr &c = *a.c;
    const dim3 grid(c.xgrid + (a.rider.kind != 0 ? 1 : 0));
    const void *valv = V8 ? (const void *)c.val8 : (const void *)c.val;
    const void *colv = C16 ? (const void *)c.col16 : (const void *)c.col;
    const XlArgs xa{c.rows, c.pw, c.cols, c.skew, c.rel16};
    if (a.e0 == nullptr)
        hipLaunchKernelGGL((k_spmv_xlw<OffT, V8, C16>), grid, dim3(XLW_BLOCK), 0, a.stream, (const OffT *)c.rowptr,
                           colv, valv, (const double *)c.dict, (const RowBlock *)c.blk, c.nblk, (const int *)c.gpid, a.x, z,
                           a.coef, a.stop, a.pin, a.npin, a.slot_out, a.skip_if_zero, a.rider, xa, a.nsc);
    else
        hipExtLaunchKernelGGL((k_spmv_xlw<OffT, V8, C16>), grid, dim3(XLW_BLOCK), 0, a.stream, a.e0, nullptr, 0,
                              (const OffT *)c.rowptr, colv, valv, (const double *)c.dict, (const RowBlock *)c.blk,
                              c.nblk, (const int *)c.gpid, a.x, z, a.coef, a.stop, a.pin, a.npin, a.slot_out,
                              a.skip_if_zero, a.rider, xa, a.nsc);
}
template <typename OffT, bool V8>
static void launch_xl(const SpmvArgs &a, double *z)
{
    if (a.c->col16 != nullptr) launch_xl_C<OffT, V8, true>(a, z);
    else launch_xl_C<OffT, V8, false>(a, z);
}

// column-swept row blocks (csb.h); K: chunks per wave and lock-step step (0: the free-running sweep)
template <typename VT, bool NARROW, int K>
static void launch_csb_K(H *h, const SpmvArgs &a)
{
    const Csr &c = *a.c;
    const VT *x = reinterpret_cast<const VT *>(a.x);
    VT *y = reinterpret_cast<VT *>(a.y);
    const int nph = std::max(c.phases, 1);
    const int ph0 = a.phase < 0 ? 0 : std::min(a.phase, nph - 1), ph1 = a.phase < 0 ? nph : ph0 + 1;
    const bool head = ph0 == 0, tail = ph1 == nph;   // this call opens / closes the product
    // max|x| first, piece by piece (what fixes the grids of the exact sums, csb.h): one pass over the vector
    // the product gathers from -- the solver's own vectors too (a bound from |x|_2 = 1 alone does not survive
    // duplicate entries, and is the looser one besides)
    const CsbPieces xpc = csb_pieces(c.cols);
    static const int tau_split = env_int("LSQRHIP_CSB_TAU", 1);
    CsbX xb{h->xmax_part, xpc.NP, tau_split, nullptr};
    if (a.xmax_in != nullptr) {   // the caller holds the piece maxima (the product that wrote x; shard_engine.h): no pass over x
        xb.xmax = a.xmax_in;
        xb.nxmax = a.nxmax_in;
        if (a.xmax_clr != nullptr && a.nxmax_in == csb_npieces(c.cols)) xb.clr = reinterpret_cast<unsigned long long *>(a.xmax_clr);
    } else if (head) {
        hipLaunchKernelGGL(k_csb_xmax<VT>, dim3(xpc.NP), dim3(VEC_BLOCK), 0, a.stream, x, (int64_t)c.cols, xpc, h->xmax_part);
    }
    // One launch per ROUND of row blocks (256 at a time, one per CU).  Every workgroup sweeps x from its
    // first to its last column; workgroups that start a sweep together stay close enough for the part of x
    // they gather from to sit in their XCD's L2, and a kernel boundary re-aligns them for the next round
    // (one launch over all blocks lets them drift apart: config 4 7.3 instead of 5.x ms).  The scalar rider
    // goes with the first launch.  LSQRHIP_CSB_ROUNDS=0: one launch.
    // With the overlap plan of the sharded engine (csb.h "Column stripes / phases") the launches fall into phases:
    // stripes -- phase k = all blocks x the J splits of part k of the gathered vector; segments -- phase k = the blocks
    // of part k of the output vector; `a.phase` picks one (the engine waits for an exchange between them).
    const int rounds = c.crounds;   // (LSQRHIP_CSB_ROUNDS at create)
    const int S = std::max(c.S, 1);
    CsbMat A{c.cval, c.cidx, c.cdel, c.ccb, c.cptr, c.crs, c.nrb, c.R, c.rows, c.cols, c.rexp, c.zcoarse, 0, 0, S, c.zsplit,
             c.cbad, std::max(c.Q, 1), c.gptr, c.NS, c.G, c.J, c.Pst, c.border, 0, S, c.cbarrier_a, c.cstagger,
             reinterpret_cast<unsigned long long *>(a.ymax_out), nullptr, reinterpret_cast<CsbHand *>(c.chand), 0, (S > 1 && c.cfuse) ? 1 : 0};
    int probe_launch = 0;   // (LSQRHIP_CSB_PROBE=1: the phase clocks of this product's first launches)
    const bool fused = A.fuse != 0;   // the last split of a block closes it inside the sweep launch: no k_csb_combine
    bool first = head;
    for (int ph = ph0; ph < ph1; ++ph) {
        int p0 = 0, p1 = c.nrb, nsp = S;   // positions of the launch order, splits per unit row
        if (c.NS > 1) {
            A.sp0 = ph * c.J;
            A.sp1 = A.sp0 + c.J;
            nsp = c.J;
        } else if (c.border != nullptr) {
            p0 = c.phase_pos[(size_t)ph];
            p1 = c.phase_pos[(size_t)ph + 1];
        }
        const int step = (rounds || nph > 1) ? std::max(1, c.grid / nsp) : std::max(c.nrb, 1);   // row blocks per launch
        for (int b0 = p0; b0 < p1 || (b0 == p0 && p0 == 0 && c.nrb == 0); b0 += step) {
            const int b1 = std::min(p1, b0 + step);
            const bool last = tail && ph == ph1 - 1 && b1 >= p1;
            A.b0 = b0;
            A.b1 = b1;
            A.hand_read = first ? 0 : 1;   // (the product's first launch derives coefficients and grids, the others read them)
            A.probe = (c.cprobe != nullptr && probe_launch < CSB_PROBE_LAUNCHES) ? c.cprobe + (size_t)probe_launch * CSB_PROBE_WGS * 8 : nullptr;
            ++probe_launch;
            Rider rider = first ? a.rider : Rider{};
            const dim3 grid(std::max(1, std::min(c.grid, (b1 - b0) * nsp)) + (rider.kind != 0 ? 1 : 0));
            hipEvent_t e0 = first ? a.e0 : nullptr, e1 = (last && (S == 1 || fused)) ? a.e1 : nullptr;
            if (e0 == nullptr && e1 == nullptr)
                hipLaunchKernelGGL((k_spmv_csb<VT, NARROW, K>), grid, dim3(CSB_BLOCK), 0, a.stream, A, x, y, a.coef, a.stop, a.pout,
                                   a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, rider, xb, a.nsc);
            else
                hipExtLaunchKernelGGL((k_spmv_csb<VT, NARROW, K>), grid, dim3(CSB_BLOCK), 0, a.stream, e0, e1, 0, A, x, y, a.coef,
                                      a.stop, a.pout, a.pin, a.npin, a.slot_in, a.slot_out, a.skip_if_zero, rider, xb,
                                      a.nsc);
            first = false;
            if (b1 >= p1) break;
        }
    }
    if (S > 1 && tail && !fused) {  // the splits' sums -> y and the blocks' partials (csb.h k_csb_combine)
        const dim3 grid(std::max(1, std::min(c.nrb * std::max(c.Q, 1), 2 * CSB_GRID)));
        if (a.e1 == nullptr)
            hipLaunchKernelGGL((k_csb_combine<VT, NARROW>), grid, dim3(CSB_BLOCK), 0, a.stream, A, x, y, a.coef, a.stop, a.pout,
                               a.pin, a.npin, a.slot_in, a.skip_if_zero, xb, a.nsc);
        else
            hipExtLaunchKernelGGL((k_csb_combine<VT, NARROW>), grid, dim3(CSB_BLOCK), 0, a.stream, nullptr, a.e1, 0, A, x, y,
                                  a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.skip_if_zero, xb, a.nsc);
    }
}

template <typename VT, bool NARROW>
static void launch_csb_N(H *h, const SpmvArgs &a)
{
    switch (a.c->clockstep) {   // (fixed at create: LSQRHIP_CSB_LOCKSTEP)
    case 0: launch_csb_K<VT, NARROW, 0>(h, a); break;
    case 1: launch_csb_K<VT, NARROW, 1>(h, a); break;
    default: launch_csb_K<VT, NARROW, 2>(h, a); break;
    }
}

template <typename VT>
static void launch_csb(H *h, const SpmvArgs &a)
{
    if (a.c->cnarrow) launch_csb_N<VT, true>(h, a);
    else launch_csb_N<VT, false>(h, a);
}

static void launch_spmv_args(H *h, const SpmvArgs &a_in)
{
    SpmvArgs a = a_in;
    a.nsc = h->nsc;
    a.f32 = h->f32;
    const Csr &c = *a.c;
    if (c.csb) {
        if (a.f32) launch_csb<float>(h, a);
        else launch_csb<double>(h, a);
        return;
    }
    if (a.f32) {  // REAL32 handle: float vectors and values (sliced ELL, row windows; never panels)
        if (c.sell == 4) {
            if (a.upd.on) launch_spat<true, float>(a);
            else launch_spat<false, float>(a);
        } else if (c.sell == 3) {
            if (a.upd.on) launch_pat<true, float>(a);
            else launch_pat<false, float>(a);
        } else if (c.sell == 2) {
            if (a.upd.on) launch_sellp<true, float>(a);
            else launch_sellp<false, float>(a);
        } else if (c.sell) {
            const int key = (c.sell_c16 ? 4 : 0) | (c.sell_v8 ? 2 : 0) | (a.upd.on ? 1 : 0);
            switch (key) {
            case 7: launch_sell_C<true, true, true, float>(a); break;
            case 6: launch_sell_C<true, true, false, float>(a); break;
            case 5: launch_sell_C<true, false, true, float>(a); break;
            case 4: launch_sell_C<true, false, false, float>(a); break;
            case 3: launch_sell_C<false, true, true, float>(a); break;
            case 2: launch_sell_C<false, true, false, float>(a); break;
            case 1: launch_sell_C<false, false, true, float>(a); break;
            default: launch_sell_C<false, false, false, float>(a); break;
            }
        } else if (h->off64) {
            launch_spmv_F<long long>(a, a.y, a.e0, a.e1);
        } else {
            launch_spmv_F<int>(a, a.y, a.e0, a.e1);
        }
        return;
    }
    if (c.sell == 4) {  // structure patterns (pat.h)
        if (a.upd.on) launch_spat<true>(a);
        else launch_spat<false>(a);
        return;
    }
    if (c.sell == 3) {  // row patterns (pat.h)
        if (a.upd.on) launch_pat<true>(a);
        else launch_pat<false>(a);
        return;
    }
    if (c.sell == 2) {  // sliced ELL, packed records (sell.h)
        if (a.upd.on) launch_sellp<true>(a);
        else launch_sellp<false>(a);
        return;
    }
    if (c.sell) {  // sliced-ELL layout (sell.h)
        const int key = (c.sell_c16 ? 4 : 0) | (c.sell_v8 ? 2 : 0) | (a.upd.on ? 1 : 0);
        switch (key) {
        case 7: launch_sell_C<true, true, true>(a); break;
        case 6: launch_sell_C<true, true, false>(a); break;
        case 5: launch_sell_C<true, false, true>(a); break;
        case 4: launch_sell_C<true, false, false>(a); break;
        case 3: launch_sell_C<false, true, true>(a); break;
        case 2: launch_sell_C<false, true, false>(a); break;
        case 1: launch_sell_C<false, false, true>(a); break;
        default: launch_sell_C<false, false, false>(a); break;
        }
        return;
    }
    if (c.P <= 1) {
        if (h->off64) launch_spmv_T<long long, false>(a, a.y, a.e0, a.e1);
        else launch_spmv_T<int, false>(a, a.y, a.e0, a.e1);
        return;
    }
    // panelled product: per-panel row sums into Z, then the combine (spmv.h "Column panels");
    // a timed launch brackets both kernels (begin of the first, end of the second)
    if (c.xlds == 2) {  // LDS-resident panels, 1024-thread workgroups (xl.h)
        const bool v8 = c.val8 != nullptr;
        if (h->off64) {
            if (v8) launch_xl<long long, true>(a, h->Z);
            else launch_xl<long long, false>(a, h->Z);
        } else {
            if (v8) launch_xl<int, true>(a, h->Z);
            else launch_xl<int, false>(a, h->Z);
        }
    } else if (h->off64) {
        launch_spmv_T<long long, true>(a, h->Z, a.e0, nullptr);
    } else {
        launch_spmv_T<int, true>(a, h->Z, a.e0, nullptr);
    }
    if (a.e1 == nullptr)
        hipLaunchKernelGGL(k_panel_combine, dim3(c.out_grid), dim3(SPMV_BLOCK), 0, a.stream, a.y, (const double *)h->Z,
                           c.rows, c.P, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in, a.skip_if_zero, a.nsc);
    else
        hipExtLaunchKernelGGL(k_panel_combine, dim3(c.out_grid), dim3(SPMV_BLOCK), 0, a.stream, nullptr, a.e1, 0, a.y,
                              (const double *)h->Z, c.rows, c.P, a.coef, a.stop, a.pout, a.pin, a.npin, a.slot_in,
                              a.skip_if_zero, a.nsc);
}

// explicit-coefficient form on the handle's stream, partials into h->partials
static void launch_spmv(H *h, const Csr &c, const double *x, double *y, const SpmvCoef *coef, const int *stop,
                        hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, bool unit_x = false)
{
    SpmvArgs a;
    a.c = &c; a.x = x; a.y = y; a.coef = coef; a.stop = stop; a.pout = h->partials; a.stream = h->stream;
    a.e0 = e0; a.e1 = e1; a.unit_x = unit_x;
    launch_spmv_args(h, a);
}

template <typename VT>
static void launch_update_T(H *h, double *pout, hipEvent_t e0, hipEvent_t e1)
{
    hipStream_t s = h->stream;
    VT *X = reinterpret_cast<VT *>(h->X), *W = reinterpret_cast<VT *>(h->W), *SE = reinterpret_cast<VT *>(h->SE);
    const VT *V = reinterpret_cast<const VT *>(h->V);
    if (e0)
        hipExtLaunchKernelGGL(k_update<VT>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, e0, e1, 0, X, W, V, SE,
                              (int64_t)h->n, (const LsqrState *)h->d_state, pout);
    else
        hipLaunchKernelGGL(k_update<VT>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, X, W, V, SE, (int64_t)h->n,
                           (const LsqrState *)h->d_state, pout);
}
static void launch_update(H *h, double *pout, hipEvent_t e0, hipEvent_t e1)
{
    if (h->f32) launch_update_T<float>(h, pout, e0, e1);
    else launch_update_T<double>(h, pout, e0, e1);
}

// ---- sequential schedule: 6 launches ----------------------------------------------------------
static void launch_iteration_seq(H *h, hipEvent_t *ev)
{
    LsqrState *st = h->d_state;
    hipStream_t s = h->stream;
    launch_spmv(h, h->A, h->V, h->U, &st->c1, &st->stop, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr, true);
    hipLaunchKernelGGL(k_s1<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->A.out_grid,
                       (const double *)nullptr, st);
    launch_spmv(h, h->AT, h->U, h->V, &st->c2, &st->stop, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr, true);
    hipLaunchKernelGGL(k_s2<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->AT.out_grid,
                       (const double *)nullptr, st);
    launch_update(h, h->partials, ev ? ev[4] : nullptr, ev ? ev[5] : nullptr);
    hipLaunchKernelGGL(k_s3<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, h->vgrid_n,
                       (const double *)nullptr, st, (const void *)h->X, h->f32 ? 1 : 0, h->d_log);
}

// ---- rider schedule -------------------------------------------------------------------------
static Rider rider_s12(H *h, int i)  // steps 1+2 of iteration i
{
    Rider r{};
    r.kind = 1;
    r.pa = h->P1[i & 1]; r.na = h->A.out_grid;
    r.pb = h->P2[i & 1]; r.nb = h->AT.out_grid;
    r.st = h->d_state;
    return r;
}
static Rider rider_s3(H *h)  // step 3 of the iteration whose K4 ran last
{
    Rider r{};
    r.kind = 2;
    r.pa = h->P3; r.na = h->vgrid_n;
    r.st = h->d_state; r.x = h->X; r.xf32 = h->f32 ? 1 : 0; r.log = h->d_log;
    return r;
}

static Rider rider_init2(H *h)  // k_s_init2 of the start of a solve (alpha, arnorm, loop entry)
{
    Rider r{};
    r.kind = 3;
    r.pa = h->P2[0]; r.na = h->AT.out_grid;
    r.st = h->d_state;
    return r;
}
// Both matrices in column-swept row blocks: each product's epilogue raises the piece maxima of the vector it writes
// (csb.h CsbMat.ymax -- the words the k_csb_xmax pass would leave) and the other product takes its grids from them: no
// pass inside the loop, same results bit for bit (LSQRHIP_CSB_XFOLD=0 at create: the passes).  Two sets per vector, by
// the parity of the iteration: the atomic maxima need zeros to start from, and the product that READS set p of a vector
// zeroes set 1 - p on its way -- between the last reader of that set and its next writer.
constexpr int MX_SET = CSB_XMAX_GRID * (VEC_BLOCK / WAVE);
static bool xmax_folded(const H *h)
{
    return h->A.csb && h->AT.csb && h->MXU != nullptr && h->MXV != nullptr;
}
// The fused schedule (pipeline 2) is used when A is neither panelled nor in column-swept row blocks.
static bool fused_schedule(const H *h) { return h->pipeline >= 2 && h->A.P <= 1 && !h->A.csb; }

// mode-1 SpMV of iteration i: alpha from the mode-2 partials of iteration i-1
// `fuse`: 1 = the launch also carries the x/w update of iteration i-1; 2 = the first launch of a solve: it carries
// w <- v / alpha instead (vec.h UpdArgs)
static void launch_k1(H *h, int i, const Rider &rider, hipEvent_t e0, hipEvent_t e1, int fuse = 0)
{
    const int par = i & 1, prev = par ^ 1;
    NormSlot *slotA = h->slots, *slotB = h->slots + 2;
    SpmvArgs a;
    a.c = &h->A; a.x = h->V; a.y = h->U; a.pin = h->P2[prev]; a.npin = h->AT.out_grid;
    a.slot_in = &slotB[prev]; a.slot_out = &slotA[par]; a.skip_if_zero = 0; a.stop = &h->d_state->stop;
    a.pout = h->P1[par]; a.rider = rider; a.stream = h->stream; a.e0 = e0; a.e1 = e1;
    a.unit_x = true;
    if (xmax_folded(h)) {   // v's piece maxima came out of the mode-2 product that wrote it; u's go to the next one
        a.xmax_in = h->MXV + par * MX_SET; a.nxmax_in = csb_npieces(h->n);
        a.xmax_clr = h->MXV + prev * MX_SET;   // (what K2(i) will raise: its last reader, K1(i-1), is done)
        a.ymax_out = h->MXU + par * MX_SET;
    }
    if (fuse) {
        UpdArgs &u = a.upd;
        u.on = fuse;
        u.par = (i - 2) & 1;  // the rotation of iteration i-1 reads rhobar2 / phibar2 of iteration i-2
        u.ugrid = h->vgrid_n;
        u.x = h->X; u.w = h->W; u.se = h->SE; u.V = h->V; u.n = h->n;
        u.st = h->d_state;
        u.alpha_prev = &slotA[prev];  // published by the mode-1 launch of iteration i-1
        u.pout = h->P3;
    }
    launch_spmv_args(h, a);
}
// mode-2 SpMV of iteration i: beta from the mode-1 partials of the same iteration
static void launch_k2(H *h, int i, const Rider &rider, hipEvent_t e0, hipEvent_t e1)
{
    const int par = i & 1;
    NormSlot *slotA = h->slots, *slotB = h->slots + 2;
    SpmvArgs a;
    a.c = &h->AT; a.x = h->U; a.y = h->V; a.pin = h->P1[par]; a.npin = h->A.out_grid;
    a.slot_in = &slotA[par]; a.slot_out = &slotB[par]; a.skip_if_zero = 1; a.stop = &h->d_state->stop;
    a.pout = h->P2[par]; a.rider = rider; a.stream = h->stream; a.e0 = e0; a.e1 = e1;
    a.unit_x = true;
    if (xmax_folded(h)) {
        a.xmax_in = h->MXU + par * MX_SET; a.nxmax_in = csb_npieces(h->m);
        a.xmax_clr = h->MXU + (par ^ 1) * MX_SET;   // (what K1(i+1) will raise)
        a.ymax_out = h->MXV + (par ^ 1) * MX_SET;   // K1(i+1) reads the set of ITS parity
    }
    launch_spmv_args(h, a);
}

// G iterations starting at global iteration i0 (1-based).  `ev`: 6 events per iteration
// (begin/end of K1, K2, K4) or nullptr.
// `solve_start`: the batch follows enqueue_solve_start directly; in the fused schedule its first mode-1 launch
// then carries k_s_init2 (as a rider) and w <- v / alpha, which enqueue_solve_start leaves out.
// `snap`: the batch ends with the state written to the pinned snapshot slots (scalar.h k_s3_snap).
static int launch_batch(H *h, int i0, int G, hipEvent_t *ev, bool solve_start = false, bool snap = false)
{
    if (!h->pipeline) {
        for (int j = 0; j < G; ++j) launch_iteration_seq(h, ev ? ev + 6 * j : nullptr);
        return LSQRHIP_OK;
    }
    LsqrState *st = h->d_state;
    hipStream_t s = h->stream;
    auto E = [&](int j, int k) -> hipEvent_t { return ev ? ev[6 * j + k] : nullptr; };
    const Rider none{};
    // pipeline 2: K4 rides inside K1 (fused update) -- two launches per iteration
    const bool fuse = fused_schedule(h);
    if (solve_start && fuse) launch_k1(h, i0, rider_init2(h), E(0, 0), E(0, 1), 2);
    else launch_k1(h, i0, none, E(0, 0), E(0, 1));
    launch_k2(h, i0, none, E(0, 2), E(0, 3));
    for (int j = 1; j < G; ++j) {
        const int i = i0 + j;
        if (fuse) {
            launch_k1(h, i, rider_s12(h, i - 1), E(j, 0), E(j, 1), 1);  // K1(i) (+) S12(i-1) (+) K4(i-1)
        } else {
            launch_k1(h, i, rider_s12(h, i - 1), E(j, 0), E(j, 1));   // K1(i)  (+) S12(i-1)
            launch_update(h, h->P3, E(j - 1, 4), E(j - 1, 5));         // K4(i-1)
        }
        launch_k2(h, i, rider_s3(h), E(j, 2), E(j, 3));            // K2(i)  (+) S3(i-1)
    }
    const int il = i0 + G - 1;  // settle the last iteration of the batch with plain kernels
    if (fuse) {
        // ... S12(il) (+) K4(il) in one launch: what K1(il + 1) would carry, without its product (scalar.h)
        const int inext = il + 1, par = inext & 1, prev = par ^ 1;
        NormSlot *slotA = h->slots, *slotB = h->slots + 2;
        UpdArgs u{};
        u.on = 1;
        u.par = (inext - 2) & 1;
        u.ugrid = h->vgrid_n;
        u.x = h->X; u.w = h->W; u.se = h->SE; u.V = h->V; u.n = h->n;
        u.st = st;
        u.alpha_prev = &slotA[prev];
        u.pout = h->P3;
        const Rider r12 = rider_s12(h, il);
        const dim3 grid(h->vgrid_n + 1);
        auto kern = h->f32 ? k_update_lazy<float> : k_update_lazy<double>;
        if (E(G - 1, 4))
            hipExtLaunchKernelGGL(kern, grid, dim3(VEC_BLOCK), 0, s, E(G - 1, 4), E(G - 1, 5), 0,
                                  (const double *)h->P2[prev], h->AT.out_grid, (const NormSlot *)&slotB[prev], u, r12,
                                  h->nsc, (const int *)&st->stop, snap ? 1 : 0);
        else
            hipLaunchKernelGGL(kern, grid, dim3(VEC_BLOCK), 0, s, (const double *)h->P2[prev], h->AT.out_grid,
                               (const NormSlot *)&slotB[prev], u, r12, h->nsc, (const int *)&st->stop, snap ? 1 : 0);
    } else {
        hipLaunchKernelGGL(k_s12, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->P1[il & 1], h->A.out_grid,
                           (const double *)h->P2[il & 1], h->AT.out_grid, st);
        launch_update(h, h->P3, E(G - 1, 4), E(G - 1, 5));
    }
    if (snap) {
        hipLaunchKernelGGL(k_s3_snap<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->P3, h->vgrid_n,
                           (const double *)nullptr, st, (const void *)h->X, h->f32 ? 1 : 0, h->d_log, h->h_state + 1);
        if (!fuse) {   // (the fused tail has left x at xout itself: spmv.h k_update_lazy copy_out)
            const int64_t bytes = (int64_t)(h->f32 ? sizeof(float) : sizeof(double)) * h->n;
            hipLaunchKernelGGL(k_out_copy, dim3(vec_grid(bytes / 8)), dim3(VEC_BLOCK), 0, s, (const LsqrState *)st,
                               (const void *)h->X, bytes);
        }
    } else
        hipLaunchKernelGGL(k_s3<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->P3, h->vgrid_n,
                           (const double *)nullptr, st, (const void *)h->X, h->f32 ? 1 : 0, h->d_log);
    return LSQRHIP_OK;
}

// The start of a solve on the handle's stream (src/lsqr.f90:242, 621-644): u = b (from the address in the pinned
// slot h->h_bslot), v = x = w = se = 0 and the Blue sums of b in one pass (k_start); beta = norm(u), u /= beta
// (k_s_init1, which also brings the initial state over from pinned host memory); v = A'u.  Then alpha = norm(v),
// v /= alpha, w = v: as k_s_init2 + k_copy_scale here, or -- fused schedule, `lean` -- carried by the first mode-1
// launch of the batch that follows (launch_batch solve_start).  The mode-2 partials land in P2[0] and
// (beta, 1/beta) in slot B[0]: exactly what the first lazy mode-1 launch (iteration 1: parity 1, previous
// parity 0) consumes.  Capturable: nothing here depends on the caller's addresses.
static int enqueue_solve_start(H *h, int wantse, bool lean)
{
    hipStream_t s = h->stream;
    const int m = h->m, n = h->n;
    LsqrState *st = h->d_state;
    // (fewer, looping workgroups for k_start -- fewer reads of the slot across PCIe -- change nothing: 128 ... 1024
    // workgroups against 1954 at config 2, profiles/r03/perf_misc.txt)
    const int sg = h->vgrid_m;
    const int g = std::max(h->vgrid_m, h->vgrid_n);
    if (h->f32)
        hipLaunchKernelGGL(k_start<float>, dim3(g), dim3(VEC_BLOCK), 0, s, (const void *const *)h->h_bslot, (float *)h->U,
                           (int64_t)m, sg, (float *)h->V, (float *)h->X, (float *)nullptr,
                           wantse ? (float *)h->SE : (float *)nullptr, (int64_t)n, h->partials);
    else
        hipLaunchKernelGGL(k_start<double>, dim3(g), dim3(VEC_BLOCK), 0, s, (const void *const *)h->h_bslot, h->U,
                           (int64_t)m, sg, h->V, h->X, (double *)nullptr, wantse ? h->SE : (double *)nullptr,
                           (int64_t)n, h->partials);
    hipLaunchKernelGGL(k_s_init1<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->partials, sg,
                       (const double *)nullptr, st, h->slots + 2, (const LsqrState *)h->h_state);
    {
        SpmvArgs a;
        a.c = &h->AT; a.x = h->U; a.y = h->V; a.coef = &st->c2; a.stop = h->d_zero; a.pout = h->P2[0]; a.stream = s;
        a.unit_x = true;
        if (xmax_folded(h)) {   // (u = b / beta: its maxima by the pass; v's for the first mode 1, iteration 1; all sets zero)
            HIPCHK(hipMemsetAsync(h->MXU, 0, sizeof(double) * 2 * MX_SET, s));
            HIPCHK(hipMemsetAsync(h->MXV, 0, sizeof(double) * 2 * MX_SET, s));
            a.ymax_out = h->MXV + 1 * MX_SET;
        }
        launch_spmv_args(h, a);
    }
    if (lean) return LSQRHIP_OK;
    hipLaunchKernelGGL(k_s_init2<true>, dim3(1), dim3(SC_BLOCK), 0, s, (const double *)h->P2[0], h->AT.out_grid,
                       (const double *)nullptr, st);
    if (h->f32)
        hipLaunchKernelGGL(k_copy_scale<float>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, (float *)h->W,
                           (const float *)h->V, (int64_t)n, (const LsqrState *)st);
    else
        hipLaunchKernelGGL(k_copy_scale<double>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->W, (const double *)h->V,
                           (int64_t)n, (const LsqrState *)st);
    return LSQRHIP_OK;
}

// Two graphs: a batch of G iterations, and the same batch preceded by the start of a solve -- a solve of
// up to G iterations is then ONE graph launch (a dozen eager calls at ~3.5 us each with the GPU idle
// behind them were 8 % of a 20-iteration solve at config 2).
// (Launching the start kernels as plain launches ahead of the first graph, so that the GPU works while the host is
// inside hipGraphLaunch, changes nothing: 565.6 against 565.9 us per 20-iteration solve, profiles/r03/perf_misc.txt.)
static int capture_graph(H *h, int G, bool with_start, int wantse, hipGraphExec_t *out)
{
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = with_start ? enqueue_solve_start(h, wantse, fused_schedule(h)) : LSQRHIP_OK;
    // (parity of iteration 1; G is even with riders.)  Graph batches are what the look-ahead poll launches: they
    // end with the snapshot.  The plain poll (poll_ahead = 0) reads slot 0 and copies the state itself.
    if (rc == LSQRHIP_OK) rc = launch_batch(h, 1, G, nullptr, with_start, h->pipeline != 0);
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    RET(rc);
    if (e != hipSuccess) return fail(LSQRHIP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(LSQRHIP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return LSQRHIP_OK;
}

static int ensure_graph(H *h, int G, int wantse)
{
    if (h->gexec && h->gexec_first && !h->graph_dirty && h->gexec_iters == G && h->gexec_pipeline == h->pipeline &&
        h->gexec_first_wantse == (wantse != 0))
        return LSQRHIP_OK;
    destroy_graph(h);
    RET(capture_graph(h, G, false, wantse, &h->gexec));
    RET(capture_graph(h, G, true, wantse, &h->gexec_first));
    h->gexec_iters = G;
    h->gexec_pipeline = h->pipeline;
    h->gexec_first_wantse = wantse != 0;
    h->graph_dirty = false;
    return LSQRHIP_OK;
}

// ---------------------------------------------------------------------------
// pieces of a solve shared by the matrix path (solve_core) and the operator path (op_api.h)
// ---------------------------------------------------------------------------
// Column splits closed by their last arriver (csb.h) keep a ticket and flags per row block, which every product returns
// to zero -- unless a product was abandoned between its launches (an engine solve that failed between two phases of a
// product: the remaining phases are never launched).  A solve must not inherit that: a few KB cleared on the handle's stream
// in front of every solve (the accumulated coarse sums zc of such an abandoned product -- used only for outliers of x --
// are NOT cleared: they cost 8 bytes per row).
static int reset_csb_tickets(H *h)
{
    for (Csr *c : {&h->A, &h->AT})
        if (c->csb && c->S > 1 && c->cfuse && c->cbad != nullptr)
            HIPCHK(hipMemsetAsync(c->cbad, 0, sizeof(int) * (size_t)c->nrb * CSB_QMAX, h->stream));
    return LSQRHIP_OK;
}

static int prepare_log(H *h, int itnlim, int want_log)
{
    if (want_log) {
        // Printable iterations (src/lsqr.f90:815-822): the first / last 10, every 10th, all of them when
        // n <= 40 -- and EVERY iteration inside the "near convergence" bands (test1 <= 10 rtol, test2 <= 10 atol,
        // test3 <= 2 ctol), which a slowly converging run can sit in for hundreds of iterations (48 x 37
        // Poisson, atol = btol = 1e-4: 272 lines of 402 iterations).  So: room for every iteration, bounded
        // at 2^20 records (112 MB); beyond that s3_step keeps the last slot for the stopping iteration and
        // raises `log_truncated`.
        const int cap = (int)std::min<int64_t>(std::max<int64_t>((int64_t)itnlim + 1, 64), 1 << 20);
        if (cap > h->log_cap) {
            if (h->d_log) (void)hipFree(h->d_log);
            h->d_log = nullptr;
            HIPCHK(hipMalloc((void **)&h->d_log, sizeof(double) * LOG_STRIDE * (size_t)cap));
            h->log_cap = cap;
            h->graph_dirty = true;
            ++h->graph_epoch;
        }
    }
    h->log_count = 0;
    return LSQRHIP_OK;
}

// initial state (src/lsqr.f90:597-617)
static int upload_initial_state(H *h, double damp, double atol, double btol, double conlim, int itnlim, int wantse,
                                int want_log, void *xout = nullptr)
{
    LsqrState init;
    std::memset(&init, 0, sizeof(init));
    init.itnlim = itnlim;
    init.damped = damp > 0.0;
    init.wantse = wantse != 0;
    init.want_log = want_log != 0;
    init.log_cap = h->log_cap;
    init.m = h->m;
    init.n = h->n;
    init.damp = damp;
    init.atol = atol;
    init.btol = btol;
    init.ctol = conlim > 0.0 ? 1.0 / conlim : 0.0;
    init.cs2 = -1.0;
    init.su = init.sv = 1.0;
    init.ns_inv = h->nsc.inv;
    init.wp32 = h->f32 ? 1 : 0;
    init.xout = xout;
    init.c1.skip = init.c2.skip = init.c2p.skip = 1;
    *h->h_state = init;   // enqueue_solve_start (or the caller) copies it to the device
    return LSQRHIP_OK;
}

// epilogue: se (:857-865), istop 2 -> 3 (:871), outputs.  h->h_state holds the settled state.
// The copies out (and k_se_finish): everything of the epilogue that needs no host knowledge of the final state.
// Enqueued behind the loop -- or, when a batch is the last one a solve can have, right behind that batch, before
// the host has seen its outcome: kernels past the stop change nothing, so x and se are final either way, and
// the host then waits once instead of twice.
static int enqueue_outputs(H *h, int wantse, double *x, double *se, bool out_on_device)
{
    hipStream_t s = h->stream;
    const int n = h->n;
    const size_t esz = h->f32 ? sizeof(float) : sizeof(double);   // x, se: float arrays for a REAL32 handle
    if (wantse && n > 0) {
        if (h->f32)
            hipLaunchKernelGGL(k_se_finish<float>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, (float *)h->SE, (int64_t)n,
                               (const LsqrState *)h->d_state);
        else
            hipLaunchKernelGGL(k_se_finish<double>, dim3(h->vgrid_n), dim3(VEC_BLOCK), 0, s, h->SE, (int64_t)n,
                               (const LsqrState *)h->d_state);
    }
    if (out_on_device) {  // a kernel of our own: starts ~6 us sooner behind the loop than a copy command does
        const int g = vec_grid((int64_t)(esz * (size_t)n / 8));
        if (n > 0)
            hipLaunchKernelGGL(k_copy_bytes, dim3(g), dim3(VEC_BLOCK), 0, s, (const void *)h->X, (void *)x,
                               (int64_t)(esz * (size_t)n));
        if (wantse && n > 0)
            hipLaunchKernelGGL(k_copy_bytes, dim3(g), dim3(VEC_BLOCK), 0, s, (const void *)h->SE, (void *)se,
                               (int64_t)(esz * (size_t)n));
        HIPCHK(hipGetLastError());
        return LSQRHIP_OK;
    }
    if (n > 0) HIPCHK(hipMemcpyAsync(x, h->X, esz * (size_t)n, hipMemcpyDeviceToHost, s));
    if (wantse && n > 0) HIPCHK(hipMemcpyAsync(se, h->SE, esz * (size_t)n, hipMemcpyDeviceToHost, s));
    return LSQRHIP_OK;
}

// `outputs_done`: 0 = not enqueued yet; 1 = enqueued, the stream may still be running them; 2 = enqueued AND the
// host has already waited for them.
static int finish_solve(H *h, int wantse, int want_log, double *x, double *se, bool out_on_device, int *istop,
                        int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm,
                        bool timed, std::chrono::steady_clock::time_point t_host0, int outputs_done = 0)
{
    hipStream_t s = h->stream;
    lsqrhip_timing_t &tm = h->timing;
    if (!outputs_done) RET(enqueue_outputs(h, wantse, x, se, out_on_device));
    const LsqrState &r = *h->h_state;
    bool need_sync = outputs_done != 2;
    if (want_log && r.itn > 0) {
        need_sync = true;
        h->log_count = std::min(r.log_count, h->log_cap);
        h->h_log.resize((size_t)h->log_count * LOG_STRIDE);
        HIPCHK(hipMemcpyAsync(h->h_log.data(), h->d_log, sizeof(double) * h->h_log.size(), hipMemcpyDeviceToHost, s));
    }
    if (need_sync) HIPCHK(hipStreamSynchronize(s));
    float loop_ms = -1.f;   // (-1 when the loop was not bracketed by events, option loop_events: a consumer that forgot to
                            // ask for them reads a time that cannot be one instead of a plausible 0)
    if (h->loop_bracketed) (void)hipEventElapsedTime(&loop_ms, h->ev_loop0, h->ev_loop1);
    tm.loop_ms = loop_ms;
    tm.itn = r.itn;
    if (!timed) tm.spmv1_launches = tm.spmv2_launches = tm.update_launches = r.itn;

    int is = r.istop;
    if (r.damped && is == 2) is = 3;
    *istop = is;
    if (itn) *itn = r.itn;
    if (anorm) *anorm = r.anorm;
    if (acond) *acond = r.acond;
    if (rnorm) *rnorm = r.rnorm;
    if (arnorm) *arnorm = r.arnorm;
    if (xnorm) *xnorm = r.xnorm;
    tm.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count();
    return LSQRHIP_OK;
}

static int solve_op(H *h, const double *b, bool b_on_device, double damp, double atol, double btol, double conlim,
                    int itnlim, int wantse, int want_log, double *x, double *se, bool out_on_device, int *istop,
                    int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm);

// ---------------------------------------------------------------------------
// solve
// ---------------------------------------------------------------------------
static int solve_core(H *h, const double *b, bool b_on_device, double damp, double atol, double btol, double conlim,
                      int itnlim, int wantse, int want_log, double *x, double *se, bool out_on_device, int *istop,
                      int *itn, double *anorm, double *acond, double *rnorm, double *arnorm, double *xnorm)
{
    if (!h) return fail(LSQRHIP_ERR_NOT_INIT, lsqrhip_error_string(LSQRHIP_ERR_NOT_INIT));
    if (h->group) {  // a system sharded over several GPUs by this process (shard_engine.h)
        if (b_on_device || out_on_device)
            return fail(LSQRHIP_ERR_ARG, "a sharded handle takes host vectors (its devices each hold a row block)");
        return solve_group_host(h, b, damp, atol, btol, conlim, itnlim, wantse, want_log, x, se, istop, itn, anorm, acond,
                                rnorm, arnorm, xnorm);
    }
    if (!istop || (!x && h->n > 0) || (!b && h->m > 0)) return fail(LSQRHIP_ERR_ARG, "null b, x or istop");
    if (wantse && !se) return fail(LSQRHIP_ERR_ARG, "wantse set but se is null");
    HIPCHK(hipSetDevice(h->device));
    if (h->op)  // user device operator (op_api.h)
        return solve_op(h, b, b_on_device, damp, atol, btol, conlim, itnlim, wantse, want_log, x, se, out_on_device,
                        istop, itn, anorm, acond, rnorm, arnorm, xnorm);
    const auto t_host0 = std::chrono::steady_clock::now();
    hipStream_t s = h->stream;
    const int m = h->m, n = h->n;
    LsqrState *st = h->d_state;

    int G = std::max(1, h->graph_iters);
    if (h->pipeline) G = (G + 1) & ~1;  // even: parity-consistent batches
    const bool timed = h->time_kernels != 0;
    const bool fused_update = fused_schedule(h);
    const bool graph = h->use_graph != 0 && !timed;
    // x of a device-resident solve is copied out by the batch that raises the stop flag (vec.h k_out_copy)
    const bool dev_out = graph && h->poll_ahead && h->pipeline != 0 && out_on_device && !wantse && h->n > 0;
    RET(prepare_log(h, itnlim, want_log));
    RET(reset_csb_tickets(h));
    RET(upload_initial_state(h, damp, atol, btol, conlim, itnlim, wantse, want_log, dev_out ? (void *)x : nullptr));

    if (graph) RET(ensure_graph(h, G, wantse));   // before anything of THIS solve is enqueued (capture)

    // u = b (solve_ez :242): k_start reads b through the pinned slot -- a device b where it lies, a host b after
    // its upload into U; then the start of the solve (enqueue_solve_start), eagerly or as the head of the first graph
    if (m > 0 && !b_on_device)
        HIPCHK(hipMemcpyAsync(h->U, b, (h->f32 ? sizeof(float) : sizeof(double)) * (size_t)m, hipMemcpyHostToDevice, s));
    *h->h_bslot = (m > 0 && b_on_device) ? (const void *)b : (const void *)h->U;
    if (!graph) {
        RET(enqueue_solve_start(h, wantse, fused_schedule(h)));
        HIPCHK(hipGetLastError());
    }

    // ---- the loop (src/lsqr.f90:673-852) ------------------------------------
    lsqrhip_timing_t &tm = h->timing;
    tm = lsqrhip_timing_t{};
    const int P = h->off64 ? 8 : 4;
    tm.spmv1_bytes = 12 * h->nnz + (int64_t)P * (m + 1) + 8ll * n + 16ll * m;
    tm.spmv2_bytes = 12 * h->nnz + (int64_t)P * (n + 1) + 8ll * m + 16ll * n;
    tm.vec_bytes = 40ll * n + (wantse ? 16ll * n : 0);

    if (timed && (int)h->ev.size() < 6 * G) {
        const size_t old = h->ev.size();
        h->ev.resize(6 * (size_t)G);
        for (size_t k = old; k < h->ev.size(); ++k) HIPCHK(hipEventCreate(&h->ev[k]));
    }
    // HIP events around the loop (timing.loop_ms) only on request: the two records cost a 20-iteration solve at
    // config 2 8-9 us of its 566 (profiles/r03/perf_misc.txt)
    const bool loop_events = h->loop_events != 0 || timed;
    h->loop_bracketed = loop_events;
    if (loop_events) HIPCHK(hipEventRecord(h->ev_loop0, s));
    // S3 raises `stop` at itn == itnlim at the latest; anything beyond this many batches
    // means the device loop is not advancing (never spin on a dead stream).
    const int64_t max_batches = (int64_t)std::max(itnlim, 0) / G + 2;
    if (graph && h->poll_ahead) {
        // Look-ahead poll: batch k+1 is enqueued BEFORE the host waits for batch k's snapshot, so
        // the wait, the check and the next graph launch (~80 us per batch of 50 at config 2)
        // overlap device work.  Never beyond itnlim; after a stop inside batch k the kernels of
        // batch k+1 return at their first instruction (stop flag) and change nothing.
        int64_t outputs_in = -1;   // the batch the outputs were enqueued behind (enqueue_outputs), if any
        // spin on the snapshot's seal instead of sleeping on the batch's event: only where the snapshot kernel is the
        // batch's LAST node -- the fused schedule, whose tail update has already left x at xout.  (In the other pipelined
        // schedules k_out_copy follows the snapshot: a seal seen there says nothing about x, so those wait for the
        // batch's event, which covers the copy.)  Also: the outputs need no later command of this stream (dev_out) and
        // nobody asked for the log (copied out behind the loop).  LSQRHIP_SPIN_POLL=0: never.
        static const int spin_env = env_int("LSQRHIP_SPIN_POLL", 1);
        const bool spin_poll = spin_env != 0 && h->pipeline != 0 && fused_update && dev_out && !want_log && !loop_events;
        bool loop1 = false;
        auto enqueue = [&](int64_t k) -> int {
            if (spin_poll) {   // (the slot's seal down before the batch that will raise it is launched)
                std::atomic_thread_fence(std::memory_order_seq_cst);
                reinterpret_cast<volatile int &>(h->h_state[1 + (k & 1)].seal) = 0;
                std::atomic_thread_fence(std::memory_order_seq_cst);
            }
            HIPCHK(hipGraphLaunch(k == 0 ? h->gexec_first : h->gexec, s));
            // the snapshot in h_state[1 + (k & 1)]: written by the batch's last kernel itself (k_s3_snap), or copied
            if (!h->pipeline)
                HIPCHK(hipMemcpyAsync(h->h_state + 1 + (k & 1), st, sizeof(LsqrState), hipMemcpyDeviceToHost, s));
            if ((k + 1) * G >= (int64_t)itnlim) {   // no batch can follow this one
                if (loop_events) HIPCHK(hipEventRecord(h->ev_loop1, s));
                loop1 = true;
                if (!dev_out) {
                    RET(enqueue_outputs(h, wantse, x, se, out_on_device));
                    outputs_in = k;
                }
            }
            HIPCHK(hipEventRecord(h->ev_batch[k & 1], s));
            return LSQRHIP_OK;
        };
        RET(enqueue(0));
        int64_t stopped_in = 0;
        for (int64_t batch = 0;; ++batch) {
            if (batch > max_batches)
                return fail(LSQRHIP_ERR_HIP, "iteration loop did not terminate (device state not advancing)");
            if ((batch + 1) * G < (int64_t)itnlim) RET(enqueue(batch + 1));
            // Wait for batch `batch`.  Its last kernel (k_s3_snap) writes the settled state into the pinned slot and
            // seals it; the host spins on the seal for up to 2 ms -- a short solve is over by then, and the spin sees
            // it ~10 us before hipEventSynchronize would have woken (profiles/r05/k20_timeline.txt: 13 us from the
            // last kernel's end to the return of the wait) -- then sleeps on the batch's event as before.
            bool sealed = false;
            if (spin_poll) {
                const volatile int *seal = &h->h_state[1 + (batch & 1)].seal;
                const int want = (int)(batch + 1);
                const auto t_spin = std::chrono::steady_clock::now();
                for (int it = 0;; ++it) {
                    if (*seal == want) {
                        sealed = true;
                        break;
                    }
                    __builtin_ia32_pause();
                    if ((it & 1023) == 1023 &&
                        std::chrono::steady_clock::now() - t_spin > std::chrono::microseconds(2000))
                        break;
                }
                std::atomic_thread_fence(std::memory_order_acquire);
            }
            if (!sealed) HIPCHK(hipEventSynchronize(h->ev_batch[batch & 1]));
            stopped_in = batch;
            if (h->h_state[1 + (batch & 1)].stop != 0) break;
            if ((batch + 1) * G >= (int64_t)itnlim)  // S3 stops at itnlim at the latest
                return fail(LSQRHIP_ERR_HIP, "iteration loop did not terminate (device state not advancing)");
        }
        // the settled state: the snapshot taken behind the stopping batch (a batch enqueued past the stop
        // returns at its stop-flag tests and has not touched it)
        *h->h_state = h->h_state[1 + (stopped_in & 1)];
        if (!loop1 && loop_events) HIPCHK(hipEventRecord(h->ev_loop1, s));
        if (dev_out)   // x was copied by the batch whose snapshot showed the stop, and that batch has been waited for --
                       // its event (which covers k_out_copy), or its seal where the snapshot is its last node (fused
                       // schedule); a batch enqueued behind it (look-ahead) repeats the copy: wait for that one too
            return finish_solve(h, wantse, want_log, x, se, out_on_device, istop, itn, anorm, acond, rnorm, arnorm,
                                xnorm, timed, t_host0, (stopped_in + 1) * G < (int64_t)itnlim ? 1 : 2);
        return finish_solve(h, wantse, want_log, x, se, out_on_device, istop, itn, anorm, acond, rnorm, arnorm, xnorm,
                            timed, t_host0, outputs_in < 0 ? 0 : (outputs_in == stopped_in ? 2 : 1));
    } else {
        for (int64_t batch = 0;; ++batch) {
            if (batch > max_batches)
                return fail(LSQRHIP_ERR_HIP, "iteration loop did not terminate (device state not advancing)");
            if (graph) {
                HIPCHK(hipGraphLaunch(batch == 0 ? h->gexec_first : h->gexec, s));
            } else {
                RET(launch_batch(h, 1 + (int)(batch * G), G, timed ? h->ev.data() : nullptr, batch == 0));
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipMemcpyAsync(h->h_state, st, sizeof(LsqrState), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (timed) {
                // launches after the stop flag was raised are no-ops: count only live ones
                const int live = std::min(G, h->h_state->itn - (int)tm.spmv1_launches);
                for (int k = 0; k < live; ++k) {
                    float a = 0, c = 0, d = 0;
                    (void)hipEventElapsedTime(&a, h->ev[6 * k + 0], h->ev[6 * k + 1]);
                    (void)hipEventElapsedTime(&c, h->ev[6 * k + 2], h->ev[6 * k + 3]);
                    // fused schedule: only the batch's last update is a kernel of its own
                    const bool own_update = !fused_update || k == G - 1;
                    if (own_update) {
                        (void)hipEventElapsedTime(&d, h->ev[6 * k + 4], h->ev[6 * k + 5]);
                        tm.update_launches += 1;
                    }
                    tm.spmv1_ms += a;
                    tm.spmv2_ms += c;
                    tm.update_ms += d;
                }
                tm.spmv1_launches += live;
                tm.spmv2_launches += live;
            }
            if (h->h_state->stop != 0) break;
        }
    }
    if (loop_events) HIPCHK(hipEventRecord(h->ev_loop1, s));

    return finish_solve(h, wantse, want_log, x, se, out_on_device, istop, itn, anorm, acond, rnorm, arnorm, xnorm,
                        timed, t_host0);
}
