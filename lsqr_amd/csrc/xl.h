// xl.h -- panelled product with LDS-resident x slices and WAVE-sized windows.
//
// With the panel's slice of x in LDS the gathers are free (scripts/gather_roof.hip: 492 G
// nonzeros/s = 5.9 TB/s of (val, col) stream with LDS gathers, against 181 G/s when they go
// to L2), so what is left to lose is the window machinery itself: in k_spmv_fused<XL> a
// 256-thread workgroup alternates streaming, staging, barrier, row sums, barrier, and with
// 72 KB of LDS only two of them fit a CU -- 21 ms per product at config 3's literal size, and
// four sub-groups in lockstep inside one 1024-thread workgroup are no faster (they share the
// barriers, so the memory pipe idles just the same).
//
// Here ONE 1024-thread workgroup per CU owns the 56 KB slice and each of its 16 WAVES runs
// windows of its own (XLW_C = 512 work units, built with that window size) with no workgroup
// barrier at all: a wave streams its window (up to 8 nonzeros in flight per lane), gathers from
// the slice, stages the products in its private 6 KB of LDS (in-order LDS queue: no barrier
// needed inside a wave), forms the row sums and moves on.  Sixteen independent streams per CU
// hide each other's phases.  The only workgroup barrier is at a change of panel.  Columns and row
// bounds are 16-bit, relative to the window (k_xl_col16, k_xl_rel16): 10 bytes per nonzero and 2
// per (panel, row).  Config 3 literal: 9.5-10 ms per product.
//
// A trip's slice is the panel of the FIRST non-empty window of its 16; entries of a window that
// reaches into the next panel are gathered from global memory (gx).  Output z[v] = raw sum of
// the (panel, row) segment, combined by k_panel_combine (spmv.h).
#pragma once

#include <type_traits>

#include "common.h"
#include "scalar.h"
#include "spmv.h"
#include "state.h"
#include "valdict.h"

namespace lsqrhip {

constexpr int XLW_BLOCK = 1024;
constexpr int XLW_WAVES = XLW_BLOCK / WAVE;  // 16 windows per trip
constexpr int XLW_C = 512;                   // window size in work units (nonzeros + rows); 256 / 320 / 384 /
                                             // 512: 11.8 / 10.9 / 10.2 / 9.9 ms per product at config 3 literal
constexpr int XLW_U = 8;                     // nonzeros per lane issued a trip ahead: XLW_C = 8 * 64; the rest
                                             // of a window (< 2 * XLW_C nonzeros: a long last row) is fetched late

// A last row of XLW_LONG nonzeros or more is streamed by the wave (phase 3) instead of staged, so a
// window stages fewer than XLW_C + XLW_LONG products: that bound, not 2 * XLW_C, sizes the LDS.
constexpr int XLW_LONG = XLW_C / 2;

// 16-bit columns for the wave-window layout: relative to the first column of the panel the
// window STARTS in.  A window reaches at most into the next panel, so the offsets stay below
// 2 * pw <= 14336: always representable.  One wave per window.
__global__ __launch_bounds__(256) void k_xl_col16(const RowBlock *__restrict__ blk, int64_t nblk,
                                                  const int *__restrict__ col, int rows, int pw,
                                                  unsigned short *__restrict__ col16)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t nw = (int64_t)gridDim.x * (256 / WAVE);
    for (int64_t b = (int64_t)blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6); b < nblk; b += nw) {
        const RowBlock q = blk[b];
        if (q.r0 >= q.r1) continue;
        const int cb = (q.r0 / rows) * pw;
        for (long long k = q.p0 + lane; k < q.pend; k += WAVE) col16[k] = (unsigned short)(col[k] - cb);
    }
}

// 16-bit row bounds to go with them: rel16[v] = rowptr[v] - p0 of the window that owns virtual row v.
// Every row of a window STARTS within the window's XLW_C work units, so the offsets are < XLW_C; a
// row ends where the next one starts, the last row of a window at the descriptor's `pend`.  The
// product then reads 2 bytes per (row, panel) instead of the 8-byte row pointer (config 3 literal:
// 1.1 GB instead of 4.5 GB per product), and the row pointers are released.  One wave per window.
template <typename OffT>
__global__ __launch_bounds__(256) void k_xl_rel16(const RowBlock *__restrict__ blk, int64_t nblk,
                                                  const OffT *__restrict__ rowptr, unsigned short *__restrict__ rel16)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t nw = (int64_t)gridDim.x * (256 / WAVE);
    for (int64_t b = (int64_t)blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6); b < nblk; b += nw) {
        const RowBlock q = blk[b];
        for (int r = q.r0 + lane; r < q.r1; r += WAVE) rel16[r] = (unsigned short)((long long)rowptr[r] - q.p0);
    }
}

// gpid[g] = panel of the first non-empty window of trip g (XLW_WAVES windows), -1 if all are empty:
// the kernel then learns a trip's slice from one prefetched word instead of scanning descriptors.
__global__ __launch_bounds__(256) void k_xl_group_panel(const RowBlock *__restrict__ blk, int64_t nblk, int rows,
                                                        int64_t ngrp, int *__restrict__ gpid)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngrp) return;
    int pid = -1;
    for (int i = 0; i < XLW_WAVES && pid < 0; ++i) {
        const int64_t b = g * XLW_WAVES + i;
        if (b < nblk) {
            const RowBlock q = blk[b];
            if (q.r0 < q.r1) pid = q.r0 / rows;
        }
    }
    gpid[g] = pid;
}

// C16 = true: 16-bit window-relative columns (k_xl_col16) and row bounds (k_xl_rel16, xa.rel16; rowptr
// is then null).
template <typename OffT, bool V8, bool C16>
__global__ __launch_bounds__(XLW_BLOCK, 1) void k_spmv_xlw(
    const OffT *__restrict__ rowptr, const void *__restrict__ colv, const void *__restrict__ valv,
    const double *__restrict__ dict, const RowBlock *__restrict__ blk, int64_t nblk, const int *__restrict__ gpid,
    const double *__restrict__ x, double *__restrict__ z, const SpmvCoef *__restrict__ coef,
    const int *__restrict__ stop, const double *__restrict__ pin, int npin, NormSlot *__restrict__ slot_out,
    int skip_if_zero, Rider rider, XlArgs xa, NScale nsc)
{
    __shared__ double xs[XL_COLS];
    __shared__ double prod[XLW_WAVES][XLW_C + XLW_LONG];
    __shared__ double red[SC_BLOCK / WAVE + 1];
    __shared__ double sdict[V8 ? VD_MAX : 1];
    __shared__ double bcast;
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    // wave index as a SCALAR: window descriptors and everything derived from them (bounds, counts,
    // bases) then live in SGPRs and are fetched with scalar loads
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {  // the scalar rider is written for 256 threads: the other waves leave
        if (tid >= SC_BLOCK) return;
        run_rider(rider, red);
        return;
    }
    if (*stop != 0) return;
    const double *__restrict__ val = static_cast<const double *>(valv);
    const unsigned char *__restrict__ val8 = static_cast<const unsigned char *>(valv);
    const int *__restrict__ col = static_cast<const int *>(colv);
    const unsigned short *__restrict__ col16 = static_cast<const unsigned short *>(colv);
    if (V8 && tid < VD_MAX) sdict[tid] = dict[tid];

    double sx;
    if (pin != nullptr) {
        // the 256-thread fixed-order reduction of the other kernels, bit for bit
        double s = 0.0;
        if (tid < SC_BLOCK && npin > 0) s = strided_sum<SC_BLOCK>(pin, npin);
        s = wave_sum(s);
        if (tid < SC_BLOCK && lane == 0) red[w] = s;
        __syncthreads();
        if (tid == 0) {
            double r = 0.0;
#pragma unroll
            for (int i = 0; i < SC_BLOCK / WAVE; ++i) r += red[i];
            bcast = r;
        }
        __syncthreads();
        const double nrm = sqrt(bcast) * nsc.inv;
        if (skip_if_zero && !(nrm > 0.0)) {
            if (wg == 0 && tid == 0) {
                slot_out->nrm = nrm;
                slot_out->scale = 1.0;
            }
            return;
        }
        sx = nrm > 0.0 ? 1.0 / nrm : 1.0;
        if (wg == 0 && tid == 0) {
            slot_out->nrm = nrm;
            slot_out->scale = sx;
        }
    } else {
        if (coef->skip != 0) return;
        sx = coef->sx;
    }
    __syncthreads();  // sdict visible

    int xpid = -1, xbase = 0;
    auto gx = [&](int cj) -> double {
        const unsigned off = (unsigned)(cj - xbase);
        if (off < (unsigned)xa.pw) return xs[off];
        return x[cj] * sx;
    };
    double *__restrict__ myprod = prod[w];

    // Software pipeline per wave: the (val, col) stream of the NEXT window is issued before this
    // window's products are staged and summed; descriptors run two windows ahead.
    typedef typename std::conditional<V8, int, double>::type RawV;
    const int64_t ngrp = (nblk + XLW_WAVES - 1) / XLW_WAVES;
    const XcdRange xr = xcd_range(ngrp, nwg, wg);
    RowBlock none;
    none.p0 = none.plast = none.pend = 0;
    none.r0 = none.r1 = 0;
    auto desc = [&](int64_t grp) -> RowBlock {
        const int64_t b = grp * XLW_WAVES + w;
        return (grp < xr.end && b < nblk) ? blk[b] : none;
    };
    // lanes per row from the window's mean row length (as in spmv.h)
    auto lanes_per_row = [](int cnt, int nr) -> int {
        int G = 1;
        if (nr > 0) {
            const int avg = cnt / nr;
            while (G < WAVE && avg > 16 * G) G <<= 1;
        }
        return G;
    };
    // Everything a window needs from memory, issued one trip ahead and in the order it is
    // consumed: a wave's loads return IN ORDER, so a load issued after the next window's stream
    // would wait for that stream -- the bounds of this lane's first row therefore travel with it.
    // UNCONDITIONAL: always the same 2 + 2 * XLW_U loads (clamped indices; an empty window reads
    // element 0 / row 0 and ignores them).  With loads under `if`s the number outstanding depends
    // on the path and the compiler must then wait for far more than the previous window's data.
    auto load_head = [&](const RowBlock &q, RawV (&av)[XLW_U], int (&cv)[XLW_U], OffT &qa, OffT &qb) {
        const bool hl = (q.pend - q.plast) >= (long long)XLW_LONG;
        const int nq = q.r0 < q.r1 ? (int)((hl ? q.plast : q.pend) - q.p0) : 0;
        const int r1q = hl ? q.r1 - 1 : q.r1;
        const int Gq = lanes_per_row(nq, r1q - q.r0);
        int rf = q.r0 + lane / Gq;
        rf = rf < r1q ? rf : (r1q > q.r0 ? r1q - 1 : q.r0);  // clamped: a valid virtual row
        if (C16) {  // raw 16-bit starts of this row and the next (the next may belong to another window:
                    // process() then takes the descriptor's end instead)
            qa = (OffT)xa.rel16[rf];
            qb = (OffT)xa.rel16[rf + 1];
        } else {
            qa = rowptr[rf];
            qb = rowptr[rf + 1];
        }
        const int lastq = nq > 0 ? nq - 1 : 0;
        const OffT qp = (OffT)q.p0;
#pragma unroll
        for (int j = 0; j < XLW_U; ++j) {
            const int e = lane + j * WAVE;
            const int ke = e < lastq ? e : lastq;
            if (V8) av[j] = (RawV)val8[qp + ke];
            else av[j] = (RawV)val[qp + ke];
            cv[j] = C16 ? (int)col16[qp + ke] : col[qp + ke];
        }
    };
    RowBlock d1 = desc(xr.first), d2 = desc(xr.first + xr.stride);
    // one trip's slice of x: the panel of the trip's first non-empty window (one word per trip,
    // read a trip ahead); uniform over the workgroup
    int pid_next = xr.first < xr.end ? gpid[xr.first] : -1;
    auto slice_for = [&](int64_t grp) {
        const int pid = pid_next;
        pid_next = grp + xr.stride < xr.end ? gpid[grp + xr.stride] : -1;
        if (pid >= 0 && pid != xpid) {
            __syncthreads();  // every wave has finished gathering from the old slice
            xpid = pid;
            xbase = pid * xa.pw;
            for (int i = tid; i < xa.pw; i += XLW_BLOCK) {
                const int cx = xbase + i;
                xs[i] = cx < xa.ncols ? x[cx] * sx : 0.0;
            }
            __syncthreads();
        }
    };
    // one window, its stream (av, cv) and first row bounds (q0, q1) loaded a trip ago: no
    // workgroup barrier in here
    auto process = [&](const RowBlock &cur, const RawV (&av)[XLW_U], const int (&cv)[XLW_U], OffT q0, OffT q1) {
        const int r0 = cur.r0, r1 = cur.r1;
        if (r0 >= r1) return;
        const OffT p0 = (OffT)cur.p0, plast = (OffT)cur.plast, pend = (OffT)cur.pend;
        const bool has_long = (pend - plast) >= (OffT)XLW_LONG;
        const int r1s = has_long ? r1 - 1 : r1;
        const int cnt = (int)((has_long ? plast : pend) - p0);  // < XLW_C + XLW_LONG
        const int nr = r1s - r0;
        const int G = lanes_per_row(cnt, nr);
        const int gl = lane & (G - 1), gid = lane / G, ngroups = WAVE / G;
        const int cb = C16 ? (r0 / xa.rows) * xa.pw : 0;  // first column of the window's own panel
        const int rfirst = r0 + gid;
        const bool have_row = rfirst < r1s;  // its bounds q0, q1 arrived with the stream
        // phase 1: stage the products of the window
#pragma unroll
        for (int j = 0; j < XLW_U; ++j) {
            const int e = lane + j * WAVE;
            const double a = V8 ? sdict[(int)av[j]] : (double)av[j];
            if (e < cnt) myprod[e] = a * gx(cb + cv[j]);
        }
        // A window holds XLW_C units = nonzeros + rows, so only one whose last row runs long has
        // more than XLW_U * 64 nonzeros (config 3: 224 per window): those few are fetched here,
        // un-pipelined, instead of doubling the loads every window issues ahead (half of which
        // were clamped duplicates).
        for (int e = lane + XLW_U * WAVE; e < cnt; e += WAVE) {
            const double a = V8 ? sdict[val8[p0 + e]] : val[p0 + e];
            const int c = C16 ? (int)col16[p0 + e] : col[p0 + e];
            myprod[e] = a * gx(cb + c);
        }
        __builtin_amdgcn_wave_barrier();  // same wave, in-order LDS queue: the products are visible
        // phase 2: row sums out of this wave's products
        if (have_row) {
            int r = rfirst;
            const int wend = (int)(pend - p0);  // where the window's last row ends
            for (;;) {
                const int s0 = C16 ? (int)q0 : (int)(q0 - p0);
                const int s1 = C16 ? (r + 1 < r1 ? (int)q1 : wend) : (int)(q1 - p0);
                double sum = 0.0;
                for (int k = s0 + gl; k < s1; k += G) sum = sum + myprod[k];
                for (int off = G >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off, WAVE);
                if (gl == 0) z[r] = sum;
                r += ngroups;
                if (r >= r1s) break;
                if (C16) {
                    q0 = (OffT)xa.rel16[r];
                    q1 = (OffT)xa.rel16[r + 1];
                } else {
                    q0 = rowptr[r];
                    q1 = rowptr[r + 1];
                }
            }
        }
        // phase 3: a long last row, split across the wave
        if (has_long) {
            const OffT len = pend - plast;
            double sum = 0.0;
            for (OffT k = lane; k < len; k += 4 * WAVE) {
                double al[4], xl[4];
                int cl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const OffT e = k + j * WAVE;
                    const OffT ke = e < len ? e : len - 1;
                    al[j] = V8 ? sdict[val8[plast + ke]] : val[plast + ke];
                    cl[j] = C16 ? cb + (int)col16[plast + ke] : col[plast + ke];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) xl[j] = gx(cl[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const OffT e = k + j * WAVE;
                    if (e < len) sum = sum + al[j] * xl[j];
                }
            }
            sum = wave_sum(sum);
            if (lane == 0) z[r1 - 1] = sum;
        }
        __builtin_amdgcn_wave_barrier();  // the next window of this wave rewrites its products
    };

    // Two register sets, A and B, used alternately (the loop is unrolled by two trips by hand: with
    // a single set copied at the end of each trip the compiler folds the copy away and waits for
    // the loads it has just issued).  In each half: issue the OTHER set's loads for the window
    // after this one, then work on this set, whose loads were issued a whole trip ago.
    RawV aA[XLW_U], aB[XLW_U];
    int cA[XLW_U], cB[XLW_U];
#pragma unroll
    for (int j = 0; j < XLW_U; ++j) {
        aA[j] = aB[j] = 0;
        cA[j] = cB[j] = 0;
    }
    OffT qA0 = 0, qA1 = 0, qB0 = 0, qB1 = 0;
    load_head(d1, aA, cA, qA0, qA1);
    for (int64_t grp = xr.first; grp < xr.end; grp += 2 * xr.stride) {
        {
            slice_for(grp);
            const RowBlock cur = d1;
            d1 = d2;
            d2 = desc(grp + 2 * xr.stride);
            load_head(d1, aB, cB, qB0, qB1);
            process(cur, aA, cA, qA0, qA1);
        }
        if (grp + xr.stride < xr.end) {  // uniform
            slice_for(grp + xr.stride);
            const RowBlock cur = d1;
            d1 = d2;
            d2 = desc(grp + 3 * xr.stride);
            load_head(d1, aA, cA, qA0, qA1);
            process(cur, aB, cB, qB0, qB1);
        }
    }
}

}  // namespace lsqrhip
