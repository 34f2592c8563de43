// csb.h -- "column-swept row blocks": the product for matrices whose columns are scattered over an
// x that does not fit an XCD's L2 (aprod mode 1 / mode 2 on random and power-law systems:
// BASELINE configs 3-5).
//
// Same contract as spmv.h's k_spmv_fused (reference src/lsqr.f90:166-174 / :186-194 fused with the
// dscal before and the dnrm2 after, :681-683 / :692-695):
//
//     y_i  <-  cy * (y_i * sy)  +  sum_j A_ij * (x_j * sx)        partial += (y_i * ns)^2
//
// Why.  With scattered columns every gathered x_j is an L1 miss, and a CU keeps only ~64 cache lines in
// flight: ~0.3 misses per clock from L2 (~220 cycles), a third of that from beyond (scripts/gather_roof.hip),
// and the (value, index) stream from HBM competes for the same slots.  Column PANELS (spmv.h) bring the
// gathers into L2 but pay for it with row pointers per (row, panel), per-panel row sums Z written and
// re-read, and a combine launch: 1.6x the algorithmic bytes at config 4.  Here instead:
//
//   * rows are cut into blocks of R <= 20352 rows, and the nonzeros of a block are stored SORTED BY
//     COLUMN: one 1024-thread workgroup (one per CU) sweeps x from left to right while it streams
//     its block -- 8-byte value + 4-byte (local row | local column) = 12 bytes per nonzero, the
//     algorithmic minimum, and no row pointers at all.  All workgroups sweep at the same pace, so
//     the part of x they are gathering from is in L2, and the 64 lanes of a gather touch neighbouring
//     lines: what a product costs is the number of LINES of x a block touches, i.e. it falls with
//     R d / n, the nonzeros a block holds per column -- hence R as large as the LDS allows.
//   * the block's row sums are accumulated IN LDS, one 64-bit INTEGER per row (ds_add_u64).  Floating-
//     point adds in an order nobody controls would not be reproducible; integer adds are exact and
//     order-free by nature.  Each product p is rounded once to a fixed binary grid, q = rint(p / g) is added,
//     and y's new part is (double)(sum of the q) * g: ONE more rounding.
//   * the grid is the ROW's own (round 4; r03 had one grid for the whole matrix, which left a row whose 1-norm
//     lay 2^s below the largest only 61 - s bits -- weighted least squares, badly scaled columns in mode 2):
//       - build: every row i gets e1_i with 2^e1_i > sum_j |a_ij| (an integer sum relative to the row's own
//         largest exponent: deterministic), and the stored values are a'_ij = a_ij 2^-e1_i -- exact, a power of
//         two; a matrix for which it is not (a value that is not finite, an exponent beyond +-900, an entry
//         2^-126 below its row in a REAL32 handle) declines this layout.  So sum_j |a'_ij| < 1 for every row
//         and the epilogue multiplies the row's sum by 2^e1_i (2 bytes per row, read once per product).
//       - product: with tau = 2^ef >= |x_j sx| for the columns the LDS sums take, |sum_j a'_ij x_j sx| < tau:
//         g = 2^(ef - 61), the sum fits with two bits to spare whatever the row length -- only the FINAL sum
//         has to fit, two's-complement adds wrap.  A row is off by <= k 2^-62 2^e1_i tau where the reference's
//         left-to-right sum is off by up to k 2^-53 |a_i|'|x|.
//   * tau is NOT simply max|x sx|: ONE large entry in x (a spike in v or u: a column or row of A scaled far above
//     the others) would push the grid of every row up, also of the rows that never touch it.  x is cut into <= 4096
//     PIECES (csb_pieces: aligned groups of 64..1024 elements dealt round-robin -- a function of its length alone) and
//     the product is handed each piece's maximum: by the k_csb_xmax pass over x, or -- inside the solver's loop -- by
//     the epilogue of the product that WROTE x (csb_group_max: the same words, no pass).  tau = min(max, 8..32 x the
//     MEDIAN piece maximum).  Columns with |x_j sx| >= tau ("big": none at
//     all for a vector without outliers, a handful otherwise) go to a SECOND set of integer sums on the grid of
//     max|x sx| itself, kept in HBM (zc, 8 bytes per row, global atomics: integer adds again, exact and order-free)
//     and added by the epilogue of the blocks that used them.  Both grids are functions of x and the row alone:
//     the result does not depend on the order of the adds, on R, on the launch shape, on which workgroup took
//     which block or on column splits -- bit-reproducible by construction.
//   * a big column whose product is beyond even the coarse bound or not finite (inf / NaN in x: never in a solve
//     that has not already failed) cannot enter an integer sum: the sweep leaves it out and raises a flag, and
//     the block's epilogue then reads the stream a second time, adds ONLY those products as doubles
//     and patches the rows concerned -- inf and NaN come out as IEEE addition gives them, like the
//     reference's (tests/test_gpu_csb.py::test_non_finite_and_huge_x_as_the_reference).
//   * the epilogue of a block forms y_i, the block's partial of sum (y ns)^2 (one per BLOCK, so the
//     fixed-order reduction is independent of the launch shape) and clears the accumulators.
//   * THE LAST SPLIT CLOSES THE BLOCK (round 6; column splits, S > 1).  Rounds 2-5 finished a split product with a second
//     launch (k_csb_combine: 7-10 % of a product that lives 300 us -- its own prologue, a launch boundary, 62-128 row blocks
//     on 256 CUs).  Now every split stores its exact sums to z -- write-through stores -- and draws a TICKET of the
//     block (one returning atomic add); the split whose add returns S - 1 -- whichever it is -- acquires, and runs the
//     block's epilogue from its own sums (still in LDS) + the other splits' z.  Integer adds: the sum is the same whoever
//     closes the block, so y, the block's partial of sum y^2 (ONE per block, the unsplit kernel's thread -> row mapping)
//     and the piece maxima are bit for bit those of S = 1.  No spinning: nobody waits for anybody.  A solve that stops
//     while the product is under way (some splits saw the flag, some not) is told by the ticket's high half: the last
//     arriver then only cleans up (zc, flags, ticket).  Chosen for blocks of TWO splits; from three on one closer's CU is the
//     bottleneck and k_csb_combine (below), which spreads the same bytes over the chip, stays 2-4 % ahead
//     (profiles/r06/fuse_by_splits_and_harness_noise.txt).  LSQRHIP_CSB_FUSE=0 / 1 at create forces either.
//   * LOCK STEP (round 5).  A CU's vector L1 returns data in REQUEST ORDER across all its waves: a gathered line of x
//     that L2 had ready in ~250 cycles, queued behind another wave's stream line from HBM (~900), waits for that one
//     and holds its slot meanwhile.  Rounds 2-4 let every wave run on its own (next chunk's stream requested right
//     behind this chunk's gathers -- in front of the gathers of fifteen other waves); the stream's and the gathers' times
//     then ADD UP and worse (scripts/csb_ceiling.hip on 64 CUs: 11.2 ms against 3.9 + 4.4), which round 4 took for the
//     ceiling of the access pattern.  Now the 16 waves of a workgroup move in lock step, K chunks per wave and step:
//         gathers of this step's chunks | s_barrier | stream of the next step's chunks | products + LDS adds | wait
//     -- every wave has REQUESTED its gathers before any wave requests new stream lines.  One barrier per step; no
//     arithmetic changes (integer adds do not care when they happen): config 4 3.47 -> 2.42 ms per product, one
//     rank's block 0.446 -> 0.342, config 5 0.370 -> 0.280 (profiles/r05/lockstep_ab.txt).  K and the free-running
//     form (LSQRHIP_CSB_LOCKSTEP=0) are template parameters of the kernel.
//
// Layout (built once by build_csb from the COO triplets, stable LSD radix sorts of csr_build.h):
//   block b = rows [rstart[b], rstart[b+1]): at most R rows, cut so that every block holds about the
//   same number of NONZEROS (equal sweeps: workgroups that start together must stay within ~1 % of
//   each other's column for the x they gather to be in L2 -- with equal ROW counts a square random
//   matrix's transpose, whose rows are Poisson(100) long, ran 16 % slower than the matrix itself);
//   its nonzeros sorted by column (ties: COO order), padded to whole chunks of 256 (pad = value 0
//   aimed at a dummy accumulator);
//   cptr[b] = first chunk of block b; val[k], idx[k] = lrow << 17 | (col - cbase[k / 256]);
//   cbase[c] = column of the first nonzero of chunk c.  A chunk spans < 2^17 columns or the build
//   gives up (an almost empty block: such a matrix keeps the panel layout).
//   NARROW form (round 4: 11 bytes per nonzero instead of 12; chosen whenever no two column-neighbours of a
//   segment lie more than 255 columns apart -- every configuration of BASELINE.json: ~5 columns apart): the 64
//   lanes of a wave take elements j * 64 + lane of a chunk (j = 0..3, "segments"); instead of idx the chunk holds,
//   per LANE, its four local rows as u16 (8 bytes: one load) and its four column DELTAS as u8 (4 bytes: one load)
//   -- delta of element e = col(e) - col(e - 1) inside its segment, 0 for the segment's first -- and cbase holds
//   the first column of each segment (4 per chunk, wave-uniform).  A lane's columns are the inclusive scan of the
//   deltas over the lanes before it: two DPP scans of two 16-bit fields each (a segment's deltas sum to < 2^16).
#pragma once

#include <climits>

#include "common.h"
#include "csr_build.h"
#include "scalar.h"
#include "state.h"

namespace lsqrhip {

constexpr int CSB_BLOCK = 1024;
constexpr int CSB_WAVES = CSB_BLOCK / WAVE;
constexpr int CSB_U = 4;                         // nonzeros per lane and step
constexpr int CSB_CHUNK = CSB_U * WAVE;          // 256: what one wave takes per step
constexpr int CSB_RMAX = 20352;                  // rows per block: one 8-byte accumulator each in 160 KB of LDS
constexpr int CSB_NACC = CSB_RMAX + 64;          // + the padding's dummy accumulator (index R)
static_assert(CSB_NACC * 8 + (CSB_WAVES + 2) * 8 + 16 <= 160 * 1024, "accumulators + reduction scratch fit the LDS");
constexpr int CSB_LCOL_BITS = 17;
constexpr unsigned CSB_LCOL_MASK = (1u << CSB_LCOL_BITS) - 1u;
static_assert(CSB_NACC <= (1 << (32 - CSB_LCOL_BITS)), "local rows fit the index word");
constexpr int CSB_GRID = 256;                    // one workgroup per CU
constexpr int CSB_NORM_FRAC = 32;                // build: row 1-norms as integer sums of ceil(|a| 2^(32 - emax_i))
constexpr int CSB_NO_EXP = INT_MIN;              // build: "this row has no nonzero value yet" in the rows' largest exponents
constexpr int CSB_E1_LIMIT = 900;                // build: rows whose 1-norm lies beyond 2^+-900 decline the layout
constexpr int CSB_QMAX = 4;                      // product: at most this many workgroups of k_csb_combine share a row block
constexpr int CSB_PROBE_LAUNCHES = 8, CSB_PROBE_WGS = 512;   // LSQRHIP_CSB_PROBE=1: phase clocks of the first launches of a product
constexpr int CSB_SMAX = 8;                      // column splits per row block at most (lsqrhip.hip build_csb)
constexpr int CSB_XHIST = 64;                    // product: piece maxima of x binned by their distance (in exponents) from the largest
// The (value, index) stream is read once: loaded non-temporal so that it does not push the part of x the
// XCD's workgroups are gathering from out of L2 (PMC before: 15 % of the gathers missed L2, 2.6x the
// layout's bytes fetched; config 4 4.80 -> 4.20 ms, config 3 at 100 per row 956 -> 900 us).
#ifndef CSB_NT_STREAM
#define CSB_NT_STREAM 1
#endif
#ifndef CSB_BARRIER_A
#define CSB_BARRIER_A 0   // lock step: a second barrier in front of the gathers (round 5's first form; slower)
#endif

// What the sweeps of a column-split product hand to its combine launch (round 6): the coefficients and the two grids every
// workgroup of the product derives from the same partials -- written by the first workgroup of every sweep launch (all the
// same values), read by k_csb_combine instead of deriving them once more (two reductions over up to 2048 partials and 4096
// piece maxima and a histogram: 4-5 us of a launch that lives 20).
struct CsbHand {
    double sx, sy, cy;
    int ef, ec, skip, pad;
};

struct CsbMat {
    const void *val;        // VT values (double; float for a REAL32 handle), each row scaled by 2^-rexp[row]
    const unsigned *idx;    // wide: [nchunks * 256] lrow << 17 | lcol; narrow: [nchunks * 64] pairs of words = a lane's 4 u16 rows
    const unsigned *dcol;   // narrow: [nchunks * 64] a lane's 4 u8 column deltas
    const int *cbase;       // wide: [nchunks] first column of each chunk; narrow: [nchunks * 4] of each segment
    const long long *cptr;  // [nrb + 1], in chunks
    const int *rstart;      // [nrb + 1] first row of each block (blocks are cut by NONZEROS, at most R rows each)
    int nrb, R, rows, cols; // R = the dummy accumulator's index = rows per block at most
    const short *rexp;      // [rows] e1_i: 2^e1_i > sum_j |a_ij|; the stored values are a_ij 2^-e1_i
    long long *zc;          // [rows] integer sums of the products with "big" columns on the coarse grid (header); all zero between products
    int b0, b1;  // the row blocks of THIS launch: [b0, b1)
    int S;       // column splits: S workgroups share a row block, each sweeping 1/S of its chunks (see below)
    long long *z;   // S > 1: the splits' exact integer sums, [S][rows]
    int *bad;       // S > 1: [nrb][CSB_QMAX] bit 0: a split of the block left a product out (beyond the bound / not
                    //        finite); bit 1: a split of the block added to zc -- one copy per workgroup of k_csb_combine
    int Q;          // S > 1: workgroups of k_csb_combine per row block (each takes a Q-th of its rows)
    // Column stripes / phases (shard_engine.h "overlap"; all off: NS = 1, border = null, sp0 = 0, sp1 = S):
    const long long *gptr;  // NS > 1: [nrb * NS + 1] first chunk of every (block, stripe) group -- chunks never straddle a stripe
    int NS, G, J, Pst;      // stripes = Pst slices x G parts; split sp = k * J + j sweeps the stripes q * G + k, q = j, j + J, ...
    const int *border;      // the launch order of the row blocks (position -> block), or null: natural order
    int sp0, sp1;           // the column splits of THIS launch: [sp0, sp1)
    int barrier_a;             // lock step: a second barrier in front of the gathers (the first form; LSQRHIP_CSB_BARRIER_A)
    int stagger;               // lock step: every other workgroup of an XCD starts this many x 2048 cycles late (0: together)
    unsigned long long *ymax;  // or null: the piece maxima of |y| (csb_pieces(rows): the words the k_csb_xmax pass over y would
                               // leave), raised by the epilogue with atomic max -- all zero on entry.  The NEXT product (the
                               // one that gathers from this y) takes its grids from them: no pass.
    unsigned long long *probe; // or null (LSQRHIP_CSB_PROBE=1 at create, measurement only): [workgroup][8] wall-clock ticks (100 MHz)
                               // of this launch's phases, written by thread 0 for the workgroup's first unit:
                               // 0 entry | 1 coefficients | 2 grids + clear (sweep begins) | 3 sweep done | 4 sums published + ticket
                               // | 5 epilogue done | 6 (last arriver: 1) | 7 unused
    CsbHand *hand;             // see CsbHand (or null): written by the first workgroup of a product's FIRST sweep launch ...
    int hand_read;             // ... and read by its later launches (further rounds, phases) when this is set, and by k_csb_combine
    int fuse;                  // S > 1: the split of a block that arrives last at the block's ticket runs the block's epilogue
                               // itself, from its own LDS sums + the other splits' z (no k_csb_combine launch)
};

// ---------------------------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------------------------
// packed[i] = (col-1) << 32 | i;  flags[0] |= bad row, flags[2] |= bad column, flags[1] |= not sorted by column
__global__ __launch_bounds__(256) void k_csb_pack_col(const int *__restrict__ rowk, const int *__restrict__ colk,
                                                      int64_t nnz, int rows, int cols,
                                                      unsigned long long *__restrict__ packed, int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int badr = 0, badc = 0, uns = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const int r = rowk[i];
        int c = colk[i];
        if (r < 1 || r > rows) badr = 1;
        if (c < 1 || c > cols) { badc = 1; c = 1; }
        if (i > 0 && colk[i - 1] > c) uns = 1;
        packed[i] = ((unsigned long long)(unsigned)(c - 1) << 32) | (unsigned long long)(unsigned)i;
    }
    if (badr) atomicOr(&flags[0], 1);
    if (badc) atomicOr(&flags[2], 1);
    if (uns) atomicOr(&flags[1], 1);
}

__global__ __launch_bounds__(256) void k_fill_int(int *__restrict__ a, int64_t n, int v)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}

// emax[row] = the largest exponent among the row's nonzero values (2^e > |a|, frexp's); flags[0] |= 1 when a value
// is not finite (such a matrix keeps another layout: its row sums are inf / NaN by floating-point addition there)
__global__ __launch_bounds__(256) void k_csb_rowemax(const int *__restrict__ rowk, const double *__restrict__ a, int64_t nnz,
                                                     int *__restrict__ emax, int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int nf = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const double v = fabs(a[i]);
        if (!(v < 1.0e308 * 10.0)) { nf = 1; continue; }   // inf, NaN
        if (v == 0.0) continue;
        int e = 0;
        (void)frexp(v, &e);
        atomicMax(&emax[rowk[i] - 1], e);
    }
    if (nf) atomicOr(&flags[0], 1);
}

// pos1[i] = original position of the i-th nonzero in column order; cnt[row] += 1; and the row 1-norms behind
// the rows' grids (header): n1[row] += ceil(|a| 2^(32 - emax[row])) -- an integer sum of terms <= 2^32, so the
// bound does not depend on the order of the atomics and is relative to the ROW's own largest value.
__global__ __launch_bounds__(256) void k_csb_pos(const unsigned long long *__restrict__ sorted, int64_t nnz,
                                                 const int *__restrict__ rowk, const double *__restrict__ a,
                                                 const int *__restrict__ emax,
                                                 unsigned *__restrict__ pos1, int *__restrict__ cnt,
                                                 unsigned long long *__restrict__ n1)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const unsigned p = (unsigned)(sorted[i] & 0xffffffffull);
        pos1[i] = p;
        const int r = rowk[p] - 1;
        atomicAdd(&cnt[r], 1);
        const double v = fabs(a[p]);
        if (v > 0.0 && v < 1.0e308 * 10.0)
            atomicAdd(&n1[r], (unsigned long long)ceil(ldexp(v, CSB_NORM_FRAC - emax[r])));   // exact, in (0, 2^32]
    }
}

// rexp[row] = e1 with 2^e1 > sum_j |a_ij| (0 for a row without nonzero values); flags[1] |= 1 when |e1| > CSB_E1_LIMIT
__global__ __launch_bounds__(256) void k_csb_rexp(const unsigned long long *__restrict__ n1, const int *__restrict__ emax,
                                                  int rows, short *__restrict__ rexp, int *__restrict__ flags)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const unsigned long long v = n1[r];
    int e1 = 0;
    if (v != 0ull) e1 = emax[r] - CSB_NORM_FRAC + (64 - __clzll((long long)v));   // v < 2^(64 - clz)
    if (e1 > CSB_E1_LIMIT || e1 < -CSB_E1_LIMIT) {
        atomicOr(&flags[1], 1);
        e1 = 0;
    }
    rexp[r] = (short)e1;
}

__global__ __launch_bounds__(256) void k_csb_maxint(const int *__restrict__ a, int64_t n, int *__restrict__ out)
{
    int m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) m = max(m, a[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, WAVE));
    if ((threadIdx.x & (WAVE - 1)) == 0 && m > 0) atomicMax(out, m);
}

// packed[i] = group(i-th nonzero in column order) << 32 | i;  group = block * NS + stripe: block b = rows
// [rstart[b], rstart[b+1]), stripe s = columns [scut[s], scut[s+1]) (NS = 1: no stripes, the group is the block)
__global__ __launch_bounds__(256) void k_csb_pack_rb(const int *__restrict__ rowk, const int *__restrict__ colk,
                                                     const unsigned *__restrict__ pos1,
                                                     int64_t nnz, const int *__restrict__ rstart, int nrb,
                                                     const int *__restrict__ scut, int NS,
                                                     unsigned long long *__restrict__ packed)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const unsigned p = pos1[i];
        const int r = rowk[p] - 1;
        int lo = 0, hi = nrb - 1;  // last b with rstart[b] <= r
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rstart[mid] <= r) lo = mid;
            else hi = mid - 1;
        }
        unsigned g = (unsigned)lo;
        if (NS > 1) {
            const int c = colk[p] - 1;
            int sl = 0, sh = NS - 1;  // last s with scut[s] <= c
            while (sl < sh) {
                const int mid = (sl + sh + 1) >> 1;
                if (scut[mid] <= c) sl = mid;
                else sh = mid - 1;
            }
            g = (unsigned)lo * (unsigned)NS + (unsigned)sl;
        }
        packed[i] = ((unsigned long long)g << 32) | (unsigned long long)(unsigned)i;
    }
}

// perm[j] = COO position of the j-th nonzero in (block, column) order: the block-order words point into the
// column order, whose positions point into the triplets
__global__ __launch_bounds__(256) void k_csb_compose(const unsigned long long *__restrict__ sorted2,
                                                     const unsigned *__restrict__ pos1, int64_t nnz,
                                                     unsigned *__restrict__ perm)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride)
        perm[j] = pos1[(unsigned)(sorted2[j] & 0xffffffffull)];
}

// One workgroup per chunk: element t of chunk c of block b is the (c - cptr[b]) * 256 + t -th nonzero
// of the block in column order, or padding.  The value stored is a 2^-rexp[row] (header).
// flags[3] |= 1 if a chunk spans 2^17 columns or more; flags[2] |= 1 if a scaled value is not exact (it left the
// normal range -- of binary32 when `f32`: the values of a REAL32 handle are narrowed after the build).
__global__ __launch_bounds__(CSB_CHUNK) void k_csb_fill(const unsigned *__restrict__ perm,
                                                        const int *__restrict__ rowk, const int *__restrict__ colk,
                                                        const double *__restrict__ a,
                                                        const long long *__restrict__ rbstart,
                                                        const long long *__restrict__ cptr,
                                                        const int *__restrict__ rstart, int nrb, int NS, int R,
                                                        const short *__restrict__ rexp, int f32,
                                                        double *__restrict__ val, unsigned *__restrict__ idx,
                                                        int *__restrict__ cbase, int *__restrict__ flags)
{
    __shared__ int s_b, s_cb;
    const long long c = blockIdx.x;
    if (threadIdx.x == 0) {   // (cptr, rbstart: per GROUP = block * NS + stripe; an empty group has no chunk)
        int lo = 0, hi = nrb * NS - 1;  // last g with cptr[g] <= c
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cptr[mid] <= c) lo = mid;
            else hi = mid - 1;
        }
        s_b = lo;
    }
    __syncthreads();
    const int gidx = s_b, b = gidx / NS;
    const long long e = (c - cptr[gidx]) * CSB_CHUNK + threadIdx.x;  // rank inside the group
    const long long j0 = rbstart[gidx], j1 = rbstart[gidx + 1];
    const bool real = j0 + e < j1;
    int col = 0, lrow = R;  // padding: the dummy accumulator
    double v = 0.0;
    if (real) {
        const unsigned p = perm[j0 + e];
        col = colk[p] - 1;
        const int row = rowk[p] - 1;
        lrow = row - rstart[b];
        const double v0 = a[p];
        const int ex = rexp[row];
        v = ldexp(v0, -ex);
        bool exact = ldexp(v, ex) == v0;
        if (f32) exact = exact && (double)(float)v == v;
        if (!exact) atomicOr(&flags[2], 1);
    }
    if (threadIdx.x == 0) s_cb = col;  // the first element of a chunk is never padding
    __syncthreads();
    const int cb = s_cb;
    const int lc = real ? col - cb : 0;
    if (lc < 0 || lc > (int)CSB_LCOL_MASK) atomicOr(&flags[3], 1);
    const long long k = c * CSB_CHUNK + threadIdx.x;
    val[k] = v;
    idx[k] = ((unsigned)lrow << CSB_LCOL_BITS) | ((unsigned)lc & CSB_LCOL_MASK);
    if (threadIdx.x == 0) cbase[c] = cb;
}

// The NARROW form of a chunk from its wide one (header): one workgroup of 256 threads per chunk, thread = element.
// flags[1] |= 1 when two column-neighbours of a segment are more than 255 columns apart (the matrix keeps the wide form).
__global__ __launch_bounds__(CSB_CHUNK) void k_csb_narrow(const unsigned *__restrict__ idx, const int *__restrict__ cbase,
                                                          unsigned short *__restrict__ row16,
                                                          unsigned char *__restrict__ dcol8, int *__restrict__ cbase4,
                                                          int *__restrict__ flags)
{
    __shared__ int s_col[CSB_CHUNK];
    const long long c = blockIdx.x;
    const int e = threadIdx.x, j = e >> 6, l = e & (WAVE - 1);
    const unsigned w = idx[c * CSB_CHUNK + e];
    const int col = cbase[c] + (int)(w & CSB_LCOL_MASK);   // (padding: local column 0 -- below the columns before it)
    s_col[e] = col;
    __syncthreads();
    // padding sits at the tail of a block's last chunk, aimed at the dummy accumulator: it takes the column of
    // the element before it (delta 0) -- any column of x will do, its value is 0
    int mycol = col, prev = l > 0 ? s_col[e - 1] : col;
    const bool pad = (int)(w >> CSB_LCOL_BITS) >= CSB_RMAX;
    if (pad) {
        int k = e;
        while (k > 0 && (int)(idx[c * CSB_CHUNK + k] >> CSB_LCOL_BITS) >= CSB_RMAX) --k;   // (the last real element)
        mycol = s_col[k];
        if (l > 0) {
            int kp = e - 1;
            while (kp > 0 && (int)(idx[c * CSB_CHUNK + kp] >> CSB_LCOL_BITS) >= CSB_RMAX) --kp;
            prev = s_col[kp];
        } else prev = mycol;
    }
    const int d = mycol - prev;
    if (d < 0 || d > 255) atomicOr(&flags[1], 1);
    const long long o = c * CSB_CHUNK + (long long)l * CSB_U + j;   // a lane's four elements side by side
    row16[o] = (unsigned short)(w >> CSB_LCOL_BITS);
    dcol8[o] = (unsigned char)(d & 255);
    if (l == 0) cbase4[c * CSB_U + j] = mycol;
}

// ---------------------------------------------------------------------------------------------
// product
// ---------------------------------------------------------------------------------------------
struct CsbX {
    const double *xmax;  // piece maxima of |x| (vec.h k_csb_xmax over the vector this product gathers from)
    int nxmax;
    int split;           // 0: tau = the bound on max|x sx| whatever x looks like (LSQRHIP_CSB_TAU=0, ablation)
    unsigned long long *clr;  // or null: nxmax words the sweeps of this product zero on their way -- the piece maxima (CsbMat.ymax)
                              // the NEXT product that writes x will raise (two sets by iteration parity, solve_loop.h)
};

// sx, sy, cy of this launch: explicit (coef) or lazy from the previous kernel's partials (pin) -- the
// 256-thread fixed-order reduction of the other kernels, bit for bit.  `nrm` is the lazy norm.
struct CsbCoef {
    double sx, sy, cy, nrm;
    bool skip;
};
__device__ __forceinline__ CsbCoef csb_coef(const SpmvCoef *__restrict__ coef, const double *__restrict__ pin, int npin,
                                            const NormSlot *__restrict__ slot_in, int skip_if_zero, NScale nsc,
                                            double *red)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    CsbCoef c{1.0, 1.0, 1.0, 0.0, false};
    if (pin != nullptr) {
        double s = 0.0;
        if (tid < SC_BLOCK && npin > 0) s = strided_sum<SC_BLOCK>(pin, npin);
        s = wave_sum(s);
        if (tid < SC_BLOCK && lane == 0) red[w] = s;
        __syncthreads();
        if (tid == 0) {
            double r = 0.0;
#pragma unroll
            for (int i = 0; i < SC_BLOCK / WAVE; ++i) r += red[i];
            red[CSB_WAVES] = r;
        }
        __syncthreads();
        c.nrm = sqrt(red[CSB_WAVES]) * nsc.inv;
        __syncthreads();
        c.skip = skip_if_zero && !(c.nrm > 0.0);   // mode 2 is skipped when beta == 0 (:691)
        c.sx = c.nrm > 0.0 ? 1.0 / c.nrm : 1.0;
        c.cy = -c.nrm;
        c.sy = slot_in->scale;
    } else {
        c.skip = coef->skip != 0;
        c.sx = coef->sx;
        c.sy = coef->sy;
        c.cy = coef->cy;
    }
    return c;
}

// The two grids of this launch (header) from the piece maxima of x (<= 4096: csb_pieces; left by the k_csb_xmax pass, by
// the epilogue of the product that wrote x, or -- the sharded engine's v -- gathered with the norms):
//   ec: 2^ec > max_j |x_j sx|                                      -- the coarse grid, and what "in range" means;
//   ef: tau = 2^ef, min(2^ec, 8..32 x the MEDIAN piece maximum)     -- the fine grid: columns with |x_j sx| < tau.
// The median piece: spikes in up to half the pieces leave it alone, and for a vector without outliers -- Gaussian
// entries: the largest of 10^7 is 1.4 x the median maximum of pieces of ~2400; power-law rows up to 10^4 long
// (config 5's u): 1.2 x -- tau is simply the bound on max|x sx| and no column is big.
// Every thread gets the same values, and every kernel of a product (sweeps, combine) derives them from the same
// partials: bit for bit the same grids.  `red`: CSB_WAVES doubles, `hist`: CSB_XHIST ints.
struct CsbGrid {
    int ef, ec;
};
__device__ __forceinline__ CsbGrid csb_grids(CsbX xb, double sx, double *red, int *hist)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x;
    double m = 0.0;
    for (int i = tid; i < xb.nxmax; i += nthreads) m = fmax(m, xb.xmax[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, WAVE));
    if (lane == 0) red[w] = m;
    if (tid < CSB_XHIST) hist[tid] = 0;
    __syncthreads();
    m = red[0];
    for (int i = 1; i < (nthreads >> 6); ++i) m = fmax(m, red[i]);
    int em = 0;
    const bool mfin = m > 0.0 && m < 1.0e308;
    if (mfin) (void)frexp(m, &em);
    // piece maxima by their distance from the largest, in exponents (zero pieces: the last bin)
    for (int i = tid; i < xb.nxmax; i += nthreads) {
        const double v = xb.xmax[i];
        int d = CSB_XHIST - 1;
        if (!(v < 1.0e308)) d = 0;   // (inf, NaN: with the largest)
        else if (v > 0.0 && mfin) {
            int e = 0;
            (void)frexp(v, &e);
            d = em - e;
            d = d < 0 ? 0 : (d > CSB_XHIST - 1 ? CSB_XHIST - 1 : d);
        }
        // one LDS add per distinct bin of the wave (nearly all pieces share two or three bins)
        unsigned long long todo = __ballot(1);
        while (todo != 0ull) {
            const int first = __ffsll((long long)todo) - 1;
            const int d0 = __shfl(d, first, WAVE);
            const unsigned long long same = __ballot(d == d0);
            if (lane == first) atomicAdd(&hist[d0], __popcll(same));
            todo &= ~same;
            if (d == d0) break;
        }
    }
    __syncthreads();
    // the median piece maximum: the first bin at which the running count reaches K
    const int K = 1 + xb.nxmax / 2;
    int dk = 0;
    if (xb.nxmax >= K) {
        int c = 0;
        for (int d = 0; d < CSB_XHIST; ++d) {
            c += hist[d];
            dk = d;
            if (c >= K) break;
        }
    }
    __syncthreads();   // (red, hist may be reused by the caller)
    const double bound = m * fabs(sx);
    int ex = 0;
    if (bound > 0.0 && bound < 1.0e308) (void)frexp(bound, &ex);  // bound < 2^ex
    CsbGrid g;
    g.ec = ex > 1020 ? 1020 : (ex < -960 ? -960 : ex);      // 2^ec and 2^(61 - ec) must stay finite and normal
    g.ef = xb.split ? g.ec - dk + 4 : g.ec;                   // > 8 x the median piece maximum, scaled
    g.ef = g.ef > g.ec ? g.ec : (g.ef < -960 ? -960 : g.ef);
    return g;
}

// Inclusive scan over the 64 lanes of a wave of two 16-bit fields packed in a word (neither overflows: callers
// guarantee field sums < 2^16) -- row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then the rows' totals handed on
// (row_bcast 15 to rows 1 and 3, row_bcast 31 to rows 2 and 3): 6 DPP adds, no LDS.
__device__ __forceinline__ unsigned wave_scan_u16x2(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
    return v;
}

// What a lane holds of a chunk's index stream, raw (loaded ahead of use), and its decoding into local rows and
// columns of x for the lane's four elements j * 64 + lane.
template <bool NARROW>
struct CsbRaw {
    unsigned w[NARROW ? 3 : CSB_U];   // wide: the four index words; narrow: two words of rows, one of deltas
    int base[NARROW ? CSB_U : 1];     // first column of the chunk (wide) / of each segment (narrow); wave-uniform
};
template <bool NARROW, bool NT>
__device__ __forceinline__ void csb_load_raw(const CsbMat &A, long long cc, int lane, CsbRaw<NARROW> &q)
{
    if (NARROW) {
        const long long k = cc * WAVE + lane;
        const unsigned *pr = A.idx + 2 * k;
        if (NT) {
            q.w[0] = __builtin_nontemporal_load(pr);
            q.w[1] = __builtin_nontemporal_load(pr + 1);
            q.w[2] = __builtin_nontemporal_load(&A.dcol[k]);
        } else {
            q.w[0] = pr[0];
            q.w[1] = pr[1];
            q.w[2] = A.dcol[k];
        }
#pragma unroll
        for (int j = 0; j < CSB_U; ++j) q.base[j] = A.cbase[cc * CSB_U + j];
    } else {
        const long long k = cc * CSB_CHUNK + lane;
#pragma unroll
        for (int j = 0; j < CSB_U; ++j) q.w[j] = NT ? __builtin_nontemporal_load(&A.idx[k + j * WAVE]) : A.idx[k + j * WAVE];
        q.base[0] = A.cbase[cc];
    }
}
template <bool NARROW>
__device__ __forceinline__ void csb_decode(const CsbRaw<NARROW> &q, int (&r)[CSB_U], int (&col)[CSB_U])
{
    if (NARROW) {
        const unsigned d = q.w[2];
        const unsigned s01 = wave_scan_u16x2((d & 0xffu) | ((d & 0xff00u) << 8));
        const unsigned s23 = wave_scan_u16x2(((d >> 16) & 0xffu) | ((d >> 8) & 0xff0000u));
        col[0] = q.base[0] + (int)(s01 & 0xffffu);
        col[1] = q.base[1] + (int)(s01 >> 16);
        col[2] = q.base[2] + (int)(s23 & 0xffffu);
        col[3] = q.base[3] + (int)(s23 >> 16);
        r[0] = (int)(q.w[0] & 0xffffu);
        r[1] = (int)(q.w[0] >> 16);
        r[2] = (int)(q.w[1] & 0xffffu);
        r[3] = (int)(q.w[1] >> 16);
    } else {
#pragma unroll
        for (int j = 0; j < CSB_U; ++j) {
            r[j] = (int)(q.w[j] >> CSB_LCOL_BITS);
            col[j] = q.base[0] + (int)(q.w[j] & CSB_LCOL_MASK);
        }
    }
}

// Rows that were sent a product of a big column beyond the coarse bound or not finite.  The sweep left those
// products out (an integer sum cannot hold them); here the chunks [c0, c1) of the block are read once more, ONLY those
// products are added -- as doubles, in LDS (`accd`: the block's accumulators, all zero on entry and on
// exit) -- and the rows concerned are patched in y: inf and NaN come out as IEEE addition gives them,
// which is what the reference's row sum does with them.  Reached when x holds inf / NaN or
// |x| is not what the bound was taken from; never by a healthy solve.
template <typename VT, bool NARROW>
__device__ void csb_add_outliers(const CsbMat &A, const VT *__restrict__ aval, long long c0, long long c1,
                                 const VT *__restrict__ x, double sx, double tau, double pmax, double *accd,
                                 VT *__restrict__ y, int row0, int nr, int rlo = 0)
{   // (rows [rlo, nr) of the block: a workgroup of k_csb_combine patches its own share only)
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (long long c = c0 + w; c < c1; c += CSB_WAVES) {
        CsbRaw<NARROW> q;
        csb_load_raw<NARROW, false>(A, c, lane, q);
        int r[CSB_U], col[CSB_U];
        csb_decode<NARROW>(q, r, col);
        const long long k = c * CSB_CHUNK + lane;
#pragma unroll
        for (int j = 0; j < CSB_U; ++j) {
            const double xs = (double)x[col[j]] * sx;
            const double p = (double)aval[k + j * WAVE] * xs;
            if (!(fabs(xs) < tau) && !(fabs(p) < pmax) && r[j] < nr && r[j] >= rlo) atomicAdd(&accd[r[j]], p);
        }
    }
    __syncthreads();
    for (int r = rlo + tid; r < nr; r += CSB_BLOCK) {
        const double v = accd[r];
        if (v != 0.0) {   // (true for NaN)
            y[row0 + r] = (VT)((double)y[row0 + r] + ldexp(v, (int)A.rexp[row0 + r]));
            accd[r] = 0.0;
        }
    }
    __syncthreads();
}

// the block's partial of sum (y ns)^2 from y as stored, with the epilogue's thread -> row mapping and reduction
template <typename VT>
__device__ __forceinline__ double csb_sumsq_rows(const VT *__restrict__ y, int row0, int nr, NScale nsc, int rlo = 0)
{
    double sq = 0.0;
    for (int r = rlo + threadIdx.x; r < nr; r += CSB_BLOCK) {
        const double ys = (double)y[row0 + r] * nsc.s;
        sq += ys * ys;
    }
    return sq;
}

// v = |x| -> the high word of a binary64 >= v (0: v == 0 or NaN -- a NaN is not a magnitude, the products find it
// themselves; 0x7ff00000: inf, also for the top 2^-20 of the finite range)
__device__ __forceinline__ unsigned csb_hi_up(double v)
{
    if (!(v > 0.0)) return 0u;
    const unsigned hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
    return hi >= 0x7fefffffu ? 0x7ff00000u : hi + 1u;
}
// The largest of an unsigned word over the 64 lanes, valid in lane 63: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then
// the rows' maxima handed on (row_bcast 15 / 31) -- 6 DPP operations, no LDS.
__device__ __forceinline__ unsigned wave_max_u32_l63(unsigned v)
{
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
    return v;
}
// The pieces of a vector of n elements whose maxima fix a product's grids (csb_grids): element i belongs to piece
// (i >> L) % NP -- aligned groups of 2^L elements dealt round-robin to NP <= 4096 pieces; L and NP depend on n alone
// (so the grids, and with them every bit of a product, do not depend on how the matrix was blocked):
//   L = 6 up to 2^18 elements, one more per doubling, 10 from 2^22 on.
// Whoever WRITES the vector leaves the maxima -- the k_csb_xmax pass over x, or the epilogue of the column-swept product
// that wrote it (csb_group_max: no pass) -- as the same words.
struct CsbPieces {
    int L, NP;
};
__host__ __device__ inline CsbPieces csb_pieces(long long n)
{
    int L = 6;
    while (L < 10 && (n >> (L + 12)) > 0) ++L;
    const long long groups = (n + (1ll << L) - 1) >> L;
    return CsbPieces{L, (int)(groups < 1 ? 1 : (groups > 4096 ? 4096 : groups))};
}
constexpr int CSB_XMAX_PIECES = 4096;
constexpr int CSB_GMX = 24;   // epilogue: a row block's (share's) groups kept in LDS while they fit, else straight to HBM

// The epilogue's share of that.  A wave's 64 consecutive rows, the first at global row `grow0`: `hv` = csb_hi_up(|y|) of
// the lane's row (0: no row).  The rows lie in at most two groups; a group's maximum goes to gmx[group - g_first] in LDS
// (`inlds`: the block's groups fit; csb_group_flush hands them on, one atomic per group and block) or straight to the
// piece in HBM (one per wave and step: at 10^7 rows 84 us per product -- same-line atomics at the memory side).
// All 64 lanes call.
__device__ __forceinline__ void csb_group_max(unsigned long long *__restrict__ ymax, CsbPieces pc, unsigned *gmx, bool inlds,
                                              long long g_first, long long grow0, unsigned hv)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const long long ga = grow0 >> pc.L, gb = (grow0 + WAVE - 1) >> pc.L;
    auto put = [&](long long g, unsigned v) {
        if (v == 0u) return;
        if (inlds) atomicMax(&gmx[g - g_first], v);
        else atomicMax(&ymax[g % pc.NP], (unsigned long long)v << 32);
    };
    if (ga == gb) {   // (uniform)
        const unsigned m = wave_max_u32_l63(hv);
        if (lane == WAVE - 1) put(ga, m);
        return;
    }
    const int cut = (int)((gb << pc.L) - grow0);   // lanes [0, cut): group ga
    const unsigned lo = wave_max_u32_l63(lane < cut ? hv : 0u);
    const unsigned hi = wave_max_u32_l63(lane >= cut ? hv : 0u);
    if (lane == WAVE - 1) {
        put(ga, lo);
        put(gb, hi);
    }
}
// after a barrier behind the last csb_group_max: the groups' maxima -> the pieces, gmx zero again (a barrier before its next use)
__device__ __forceinline__ void csb_group_flush(unsigned long long *__restrict__ ymax, CsbPieces pc, unsigned *gmx, bool inlds,
                                                long long g_first, int ng)
{
    const int tid = threadIdx.x;
    if (inlds && tid < ng) {
        const unsigned v = gmx[tid];
        gmx[tid] = 0u;
        if (v != 0u) atomicMax(&ymax[(g_first + tid) % pc.NP], (unsigned long long)v << 32);
    }
}
// ... the maxima from y as stored (the outlier path: y was patched after the epilogue computed it), rows [rlo, nr)
template <typename VT>
__device__ __forceinline__ void csb_group_max_rows(unsigned long long *__restrict__ ymax, CsbPieces pc, unsigned *gmx, bool inlds,
                                                   long long g_first, const VT *__restrict__ y, int row0, int nr, int rlo = 0)
{
    const int tid = threadIdx.x;
    for (int rb = rlo; rb < nr; rb += CSB_BLOCK) {
        const int r = rb + tid;
        const unsigned hv = r < nr ? csb_hi_up(fabs((double)y[row0 + r])) : 0u;
        csb_group_max(ymax, pc, gmx, inlds, g_first, (long long)row0 + rb + (tid & ~(WAVE - 1)), hv);
    }
}

// A row's sum from its integer parts: the LDS sum on the fine grid, and -- `coarse`: the block used them -- the
// HBM sum of its big columns on the coarse grid (taken and cleared with agent-scope atomics: the adds of other
// workgroups' sweeps -- column splits -- were performed at memory, not in this XCD's L2).  One rounding per
// part and one for their sum; the row's power of two is exact.
__device__ __forceinline__ double csb_row_sum(const CsbMat &A, long long fine, int row, CsbGrid gr, bool coarse,
                                              int e_known = INT_MIN)   // (the row's exponent, when the caller has loaded it)
{
    const int e = e_known != INT_MIN ? e_known : (int)A.rexp[row];
    double sum = ldexp((double)fine, gr.ef - 61 + e);
    if (coarse) {
        const long long c = (long long)__hip_atomic_load((unsigned long long *)&A.zc[row], __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
        if (c != 0) {
            sum = sum + ldexp((double)c, gr.ec - 61 + e);
            __hip_atomic_store((unsigned long long *)&A.zc[row], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return sum;
}

// ---- the last split closes the block (header) -----------------------------------------------------------------------
// bad[b * CSB_QMAX + 0]: flags of the block (bit 0: a split left a product out, bit 1: a split added to zc);
// bad[b * CSB_QMAX + 1]: the ticket -- low half: splits that have arrived, high half: how many of them had seen the stop flag.
constexpr int CSB_TICKET_STOPPED = 1 << 16;
// Write-through stores of the splits' sums (MI355X_MICROARCH.md "inter-workgroup visibility": every handed-off byte
// stored sc1, every storing wave drained, then the counter; the consumer acquires and loads plainly).
__device__ __forceinline__ void csb_store16_wt(long long *p, unsigned long long a, unsigned long long b)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void csb_store8_wt(long long *p, unsigned long long a)
{
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u v = {(unsigned)a, (unsigned)(a >> 32)};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// ONE lane, behind a workgroup barrier that follows every wave's `s_waitcnt vmcnt(0)` (its write-through z stores have
// landed; the adds to zc and to the flags are atomics, performed at memory): draw the ticket; the last arriver acquires,
// takes the block's flags and lowers flags and ticket for the next product.
// Returns 0: not the last; else 1 | flags << 1 | (a split had stopped) << 3.
__device__ __forceinline__ int csb_ticket(const CsbMat &A, int b, int myflags, bool stopped)
{
    int *fl = &A.bad[b * CSB_QMAX], *tk = &A.bad[b * CSB_QMAX + 1];
    if (myflags != 0) {
        (void)__hip_atomic_fetch_or(fl, myflags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int old = __hip_atomic_fetch_add(tk, 1 + (stopped ? CSB_TICKET_STOPPED : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old & (CSB_TICKET_STOPPED - 1)) != A.S - 1) return 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int flags = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 3;
    const bool anystopped = stopped || (old >> 16) != 0;
    __hip_atomic_store(fl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 1 | (flags << 1) | (anystopped ? 8 : 0);
}
// what the last arriver of a block does when a split of it never ran (the solve stopped under the product): y is no longer
// wanted, but zc must be all zero again for the next product of this matrix
__device__ __forceinline__ void csb_cleanup_block(const CsbMat &A, int row0, int nr, int verdict)
{
    if (verdict & 4)   // (flags bit 1: a split added to zc)
        for (int r = threadIdx.x; r < nr; r += CSB_BLOCK)
            __hip_atomic_store((unsigned long long *)&A.zc[row0 + r], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the units of a workgroup that found the stop flag up: they still draw their tickets (as "stopped"), so that every
// block's ticket comes back to zero whatever mix of splits ran
__device__ void csb_fused_stopped(const CsbMat &A, int wg, int nwg, int *s_word)
{
    const int nsp = A.sp1 - A.sp0;
    const int nunits = (A.b1 - A.b0) * nsp;
    for (int u = wg; u < nunits; u += nwg) {
        const int pos = A.b0 + u / nsp;
        const int b = A.border != nullptr ? A.border[pos] : pos;
        if (threadIdx.x == 0) *s_word = csb_ticket(A, b, 0, true);
        __syncthreads();
        const int verdict = *s_word;
        if (verdict & 1) csb_cleanup_block(A, A.rstart[b], A.rstart[b + 1] - A.rstart[b], verdict);
        __syncthreads();
    }
}
// K: chunks a wave takes per LOCK-STEP step (round 5; header "lock step"), or 0: the free-running sweep of rounds 2-4
// (every wave on its own, next chunk's stream requested behind this chunk's gathers) -- kept for the A/B
// (LSQRHIP_CSB_LOCKSTEP=0 at create).  Same sums bit for bit: integer adds do not care when they happen.
template <typename VT = double, bool NARROW = false, int K = 0>
__global__ __launch_bounds__(CSB_BLOCK, 1) void k_spmv_csb(
    CsbMat A, const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef,
    const int *__restrict__ stop, double *__restrict__ partials, const double *__restrict__ pin, int npin,
    const NormSlot *__restrict__ slot_in, NormSlot *__restrict__ slot_out, int skip_if_zero, Rider rider, CsbX xb,
    NScale nsc)
{
    __shared__ unsigned long long acc[CSB_NACC];
    __shared__ double red[CSB_WAVES + 2];
    __shared__ int s_bad, s_big;
    __shared__ unsigned gmx[CSB_GMX];
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int shift = rider.kind != 0 ? 1 : 0;
    const int nwg = (int)gridDim.x - shift;
    const int wg = (int)blockIdx.x - shift;
    if (wg < 0) {  // the scalar rider is written for 256 threads: the other waves leave
        if (tid >= SC_BLOCK) return;
        run_rider(rider, red);
        return;
    }
    const bool fused = A.S > 1 && A.fuse != 0;
    unsigned long long *pb = (A.probe != nullptr && tid == 0) ? A.probe + (size_t)wg * 8 : nullptr;
    if (pb) pb[0] = wall_clock64();
    if (*stop != 0) {
        if (fused) csb_fused_stopped(A, wg, nwg, &s_bad);
        return;
    }
    // lock step: all CUs in the same phase would use HBM and the L2s in turn -- half of each XCD's workgroups
    // (workgroup i runs on XCD i % 8) start half a step late, so that one half gathers while the other streams
    if (K > 0 && A.stagger > 0 && ((wg >> 3) & 1))
        for (int i = 0; i < A.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    const VT *__restrict__ aval = static_cast<const VT *>(A.val);
    if (xb.clr != nullptr)
        for (int i = wg * CSB_BLOCK + tid; i < xb.nxmax; i += nwg * CSB_BLOCK) xb.clr[i] = 0ull;

    // The coefficients and the two grids are the same in every workgroup of every launch of a product (same partials, same
    // piece maxima, same functions).  The product's FIRST launch derives them and its first workgroup leaves them in
    // A.hand; later launches of the product -- the second round of row blocks, the phases of the overlap plan -- and the
    // combine launch READ them (a kernel boundary lies between): two reductions over up to 2048 partials and 4096 piece
    // maxima and a histogram less, 4-5 us per launch.
    double sx, sy, cy;
    CsbGrid gr;
    if (A.hand != nullptr && A.hand_read) {
        const CsbHand hd = *A.hand;
        if (hd.skip) return;
        sx = hd.sx; sy = hd.sy; cy = hd.cy;
        gr.ef = hd.ef; gr.ec = hd.ec;
        if (pb) pb[1] = wall_clock64();
    } else {
        const CsbCoef co = csb_coef(coef, pin, npin, slot_in, skip_if_zero, nsc, red);
        if (pin != nullptr && wg == 0 && tid == 0) {
            slot_out->nrm = co.nrm;
            slot_out->scale = co.skip ? 1.0 : co.sx;
        }
        const bool handoff = A.hand != nullptr && wg == 0 && tid == 0;
        if (co.skip) {
            if (handoff) A.hand->skip = 1;
            return;
        }
        sx = co.sx; sy = co.sy; cy = co.cy;
        if (pb) pb[1] = wall_clock64();
        // the binary grids of this launch (the histogram of the piece maxima borrows the accumulators' space)
        gr = csb_grids(xb, sx, red, reinterpret_cast<int *>(acc));
        if (handoff) *A.hand = CsbHand{sx, sy, cy, gr.ef, gr.ec, 0, 0};
    }
    const double tau = ldexp(1.0, gr.ef);          // columns with |x sx| below this: the LDS sums
    const double ginv = ldexp(1.0, 61 - gr.ef);    // 1 / g of the fine grid
    const double pmax2 = ldexp(1.0, gr.ec);        // a product of a big column is in range below this
    const double ginv2 = ldexp(1.0, 61 - gr.ec);   // 1 / g of the coarse grid

    for (int i = tid; i < CSB_NACC; i += CSB_BLOCK) acc[i] = 0ull;
    if (tid == 0) {
        s_bad = 0;
        s_big = 0;
    }
    if (tid < CSB_GMX) gmx[tid] = 0u;
    const CsbPieces pc = csb_pieces(A.rows);
    __syncthreads();
    if (pb) pb[2] = wall_clock64();

    // Column splits (few rows: fewer row blocks than CUs).  S workgroups share a block, each sweeping a
    // contiguous S-th of its column-sorted chunks into accumulators of its own; their integer sums go to
    // z and k_csb_combine adds them -- exact, so the result is bit for bit what ONE workgroup would have
    // produced.  A block of R rows then still holds R d / n nonzeros per column although 256 / S blocks
    // cover the matrix: R can stay large (one rank's block of config 4 at N = 8).
    // the units of this launch: row blocks at positions [b0, b1) of the launch order x column splits [sp0, sp1)
    const int nsp = A.sp1 - A.sp0;
    const int nunits = (A.b1 - A.b0) * nsp;
    for (int u = wg; u < nunits; u += nwg) {
        const int pos = A.b0 + u / nsp, sp = A.sp0 + u % nsp;
        const int b = A.border != nullptr ? A.border[pos] : pos;
        const int row0 = A.rstart[b];
        const int nr = A.rstart[b + 1] - row0;
        // the chunk ranges of the unit: one -- an S-th of the block's chunks -- or, with column stripes, those of
        // the stripes q * G + k for q = j, j + J, ... (split sp = k * J + j: part k of the slices j, j + J, ...)
        const int kpart = A.NS > 1 ? sp / A.J : 0, jsub = A.NS > 1 ? sp % A.J : 0;
        const int nranges = A.NS > 1 ? (A.Pst - jsub + A.J - 1) / A.J : 1;
        auto range = [&](int i, long long &c0, long long &c1) {
            if (A.NS > 1) {
                const long long gi = (long long)b * A.NS + (long long)(jsub + i * A.J) * A.G + kpart;
                c0 = A.gptr[gi];
                c1 = A.gptr[gi + 1];
            } else {
                const long long cb0 = A.cptr[b], cb1 = A.cptr[b + 1];
                c0 = cb0 + ((cb1 - cb0) * sp) / A.S;
                c1 = cb0 + ((cb1 - cb0) * (sp + 1)) / A.S;
            }
        };
        bool outlier = false, tookbig = false;
      for (int ri = 0; ri < nranges; ++ri) {
        long long c0, c1;
        range(ri, c0, c1);
        // software pipeline: the (value, index) stream of the wave's NEXT chunk is in flight while the
        // gathers and the LDS adds of this one run (two register sets, loads unconditional: clamped)
        double av[CSB_U], bv[CSB_U];
        CsbRaw<NARROW> iv, jv;
        const long long clast = c1 > c0 ? c1 - 1 : c0;
        auto issue = [&](long long c, double (&a)[CSB_U], CsbRaw<NARROW> &q) {
            const long long cc = c < clast ? c : clast;
            csb_load_raw<NARROW, CSB_NT_STREAM != 0>(A, cc, lane, q);   // read-once stream: non-temporal, so that it does
            const long long k = cc * CSB_CHUNK + lane;                  // not push x out of L2
#pragma unroll
            for (int j = 0; j < CSB_U; ++j)
                a[j] = CSB_NT_STREAM ? (double)__builtin_nontemporal_load(&aval[k + j * WAVE]) : (double)aval[k + j * WAVE];
        };
        // One step of a wave, in the order the memory system wants it: (1) this chunk's columns decoded and its four
        // gathers of x issued; (2) the NEXT chunk's stream requested; (3) the products and the LDS adds.  Loads return
        // in order, so waiting for the gathers at (3) never waits for the stream requested at (2) -- the order is
        // pinned with scheduling barriers (the compiler found it by itself for the wide form, not for the narrow
        // one: its gathers were issued and waited for one by one, the prefetched stream with them).
        int r[CSB_U];
        double xv[CSB_U];
        auto gather = [&](const CsbRaw<NARROW> &q) {
            int col[CSB_U];
            csb_decode<NARROW>(q, r, col);
#pragma unroll
            for (int j = 0; j < CSB_U; ++j) xv[j] = (double)x[col[j]];
            __builtin_amdgcn_sched_barrier(0);
        };
        auto accumulate = [&](const double (&a)[CSB_U]) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CSB_U; ++j) xv[j] = xv[j] * sx;
            bool anybig = false;
#pragma unroll
            for (int j = 0; j < CSB_U; ++j) {
                const double p = a[j] * xv[j];
                const bool big = !(fabs(xv[j]) < tau);   // a big column (or x not finite): not for the fine grid
                anybig |= big;
                const long long q1 = __double2ll_rn(p * ginv);
                atomicAdd(&acc[r[j]], big ? 0ull : (unsigned long long)q1);
            }
            if (__any(anybig)) {   // (wave-uniform; never taken for a vector without outliers)
#pragma unroll
                for (int j = 0; j < CSB_U; ++j) {
                    if (!(fabs(xv[j]) < tau) && r[j] < nr) {   // (padding aims at the dummy accumulator R >= nr)
                        const double p = a[j] * xv[j];
                        if (fabs(p) < pmax2)
                            atomicAdd((unsigned long long *)&A.zc[row0 + r[j]], (unsigned long long)__double2ll_rn(p * ginv2));
                        else outlier = true;   // beyond the coarse bound, or not finite: left to the outlier pass
                        tookbig = true;
                    }
                }
            }
        };
        if (K == 0) {
            if (c0 + w < c1) {
                issue(c0 + w, av, iv);
                for (long long c = c0 + w; c < c1; c += 2 * CSB_WAVES) {
                    gather(iv);
                    issue(c + CSB_WAVES, bv, jv);
                    accumulate(av);
                    if (c + CSB_WAVES < c1) {  // uniform
                        gather(jv);
                        issue(c + 2 * CSB_WAVES, av, iv);
                        accumulate(bv);
                    }
                }
            }
        } else {
            // LOCK STEP (header): all 16 waves move together, KK chunks per wave and step --
            //   decode + gathers of this step's chunks | BARRIER | stream of the next step's chunks |
            //   products + LDS adds (behind the gathers, the stream in flight) | wait for the stream |
            // The barrier: every wave's gathers are REQUESTED (not back) before any wave requests its next stream.
            // Data returns in request order across the CU: a stream request queued behind gathers holds nobody up, a
            // gather queued behind another wave's NEW stream request waits an HBM round trip for a line L2 had ready.
            // (The first form also had a barrier in FRONT of the gathers -- "every wave's stream has landed" -- so that
            // no gather ever stood behind a stream line.  Behind the LAST stream lines of slower waves a gather waits
            // no longer than it would have at that barrier, and the L1 does not run dry in between: config 4's sweeps
            // 2.89 -> 2.50 ms in scripts/csb_break.hip, profiles/r05/csb_break_d_single_barrier.txt; in the library it is
            // slower on every shape, profiles/r05/lockstep_barrier_a.txt.  LSQRHIP_CSB_BARRIER_A=1 brings it back.)
            // The step count is the same for every wave (barriers inside); a wave without a real chunk in a step
            // loads the range's last chunk again (clamped) and adds nothing.
            constexpr int KK = K > 0 ? K : 1;
            const long long nch = c1 - c0;
            const int nsteps = (int)((nch + (long long)KK * CSB_WAVES - 1) / ((long long)KK * CSB_WAVES));
            double a0[KK][CSB_U], a1[KK][CSB_U];
            CsbRaw<NARROW> q0[KK], q1[KK];
            auto issue_set = [&](long long cfirst, double (&a)[KK][CSB_U], CsbRaw<NARROW> (&q)[KK]) {
#pragma unroll
                for (int k = 0; k < KK; ++k) issue(cfirst + (long long)k * CSB_WAVES, a[k], q[k]);
            };
            auto lockstep = [&](long long cfirst, double (&a)[KK][CSB_U], CsbRaw<NARROW> (&q)[KK], double (&an)[KK][CSB_U],
                                CsbRaw<NARROW> (&qn)[KK]) {
                int rr[KK][CSB_U];
                double xx[KK][CSB_U];
                if (CSB_BARRIER_A || A.barrier_a) __builtin_amdgcn_s_barrier();   // A (uniform; the first form: see above)
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    int col[CSB_U];
                    csb_decode<NARROW>(q[k], rr[k], col);
#pragma unroll
                    for (int j = 0; j < CSB_U; ++j) xx[k][j] = (double)x[col[j]];
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();   // B
                __builtin_amdgcn_sched_barrier(0);
                issue_set(cfirst + (long long)KK * CSB_WAVES, an, qn);
                __builtin_amdgcn_sched_barrier(0);
                // A wave without a real chunk in this step (clamped loads) adds zeros to rows of the range's last chunk:
                // by SELECT, not by a branch around the adds -- with a branch the compiler sinks that chunk's gathers
                // into it, behind the barrier and the stream requests (seen in the ISA of the first build: half of the
                // gathers went out AFTER the wave's own stream), which is the very order this loop exists to avoid.
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const bool live = cfirst + (long long)k * CSB_WAVES < c1;   // (uniform)
                    double am[CSB_U];
#pragma unroll
                    for (int j = 0; j < CSB_U; ++j) {
                        r[j] = rr[k][j];
                        xv[j] = live ? xx[k][j] : 0.0;
                        am[j] = live ? a[k][j] : 0.0;
                    }
                    accumulate(am);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next step's stream has landed
            };
            // (Round 6 tried the stream TWO steps ahead -- three register sets in rotation, no wait at a step's end, the
            // lock step's order inside a step: 1-4 % SLOWER on every shape, profiles/r06/sweep_two_steps_ahead_ab.txt.  The
            // CU's L1 returns data in request order: the gathers of step s + 1 stand behind the stream requested in step s
            // whichever step consumes it, so a step still lasts that stream's round trip.)
            if (nsteps > 0) {   // (uniform)
                issue_set(c0 + w, a0, q0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int st = 0; st < nsteps; st += 2) {
                    const long long cfirst = c0 + (long long)st * KK * CSB_WAVES + w;
                    lockstep(cfirst, a0, q0, a1, q1);
                    if (st + 1 < nsteps) lockstep(cfirst + (long long)KK * CSB_WAVES, a1, q1, a0, q0);
                }
            }
        }
      }   // (ranges of the unit)
        if (outlier) s_bad = 1;
        if (tookbig) {
            s_big = 1;
            __threadfence();   // this wave's adds to zc are performed before anyone reads them back
        }
        __syncthreads();
        if (pb && u == wg) pb[3] = wall_clock64();
        // epilogue of the block: y, its partial of sum (y ns)^2, accumulators cleared
        const bool bad = s_bad != 0, big = s_big != 0;
        __syncthreads();   // (everyone has read the flags: they may be lowered)
        int verdict = 0;   // fused splits: what the block's ticket said (csb_ticket); 0: this split is not the one that closes the block
        if (fused) {
            // the sums as they are to z, kept in LDS as well -- the last arriver uses its own.  WRITE-THROUGH stores (sc0 sc1,
            // 16 bytes: pairs of rows from an even word on): the bytes leave this XCD's L2 as they are written and the
            // ticket needs no release fence -- a `buffer_wbl2` per workgroup with 160 KB freshly dirtied each cost more than
            // the combine launch it replaces (profiles/r06/fuse_ab_writethrough.txt against the first form, +10 us)
            {
                long long *zs = A.z + (size_t)sp * A.rows;
                const int odd = (int)(((long long)sp * A.rows + row0) & 1);   // word parity of the block's first sum: pairs are 16-byte aligned
                if (tid == 0 && odd && nr > 0) csb_store8_wt(&zs[row0], acc[0]);
                const int npair = (nr - odd) >> 1;
                for (int pi = tid; pi < npair; pi += CSB_BLOCK) {
                    const int r = odd + 2 * pi;
                    csb_store16_wt(&zs[row0 + r], acc[r], acc[r + 1]);
                }
                if (tid == 0 && nr > odd && ((nr - odd) & 1)) csb_store8_wt(&zs[row0 + nr - 1], acc[nr - 1]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave, before the barrier the ticket lane waits at
            __syncthreads();
            if (tid == 0) {
                s_big = csb_ticket(A, b, (bad ? 1 : 0) | (big ? 2 : 0), false);
                s_bad = 0;
            }
            __syncthreads();
            verdict = s_big;
            __syncthreads();   // (everyone has read the word)
            if (tid == 0) s_big = 0;
            if (pb && u == wg) {
                pb[4] = wall_clock64();
                pb[6] = (unsigned long long)(verdict & 1);
            }
            if (!(verdict & 1) || (verdict & 8)) {   // not the last -- or the last of a product the solve stopped under: no epilogue
                if (verdict & 1) csb_cleanup_block(A, row0, nr, verdict);
                for (int r = tid; r < nr; r += CSB_BLOCK) acc[r] = 0ull;
                if (tid == 0) acc[A.R] = 0ull;
                __syncthreads();
                continue;
            }
            // This split closes the block.  (What bounds it is the ~64 lines one CU keeps in flight: 0.85 MB of sums, y and
            // exponents through ONE CU take 28 us for a full block at S = 4 -- as long as the combine launch took on the whole
            // chip.  Requesting every line it will read at once, ahead of the epilogue, made it 55 us: profiles/r06/
            // fuse_touch_lines_negative.txt; two register sets in turn spilled the sweep.)
        } else if (A.S > 1) {  // a split: the exact sums as they are
            long long *zs = A.z + (size_t)sp * A.rows + row0;
            for (int r = tid; r < nr; r += CSB_BLOCK) {
                __builtin_nontemporal_store((long long)acc[r], &zs[r]);   // (written once, read once by k_csb_combine)
                acc[r] = 0ull;
            }
            if (tid == 0) {
                acc[A.R] = 0ull;
                s_bad = 0;
                s_big = 0;
                if (bad || big)
                    for (int qi = 0; qi < A.Q; ++qi) atomicOr(&A.bad[b * CSB_QMAX + qi], (bad ? 1 : 0) | (big ? 2 : 0));
            }
            __syncthreads();
            continue;
        }
        // (a fused split that closes its block: the flags are the BLOCK's -- what any of its splits ran into)
        const bool bad_b = fused ? (verdict & 2) != 0 : bad, big_b = fused ? (verdict & 4) != 0 : big;
        double sq = 0.0;
        const bool pieces = A.ymax != nullptr && !bad_b;   // (uniform; with outliers: from y as patched, below)
        const long long g_first = (long long)row0 >> pc.L;
        const int ng = nr > 0 ? (int)((((long long)row0 + nr - 1) >> pc.L) - g_first) + 1 : 0;
        const bool inlds = ng <= CSB_GMX;
        // y and the rows' exponents of FOUR steps at a time are requested before any of them is used: 5 round trips to
        // memory for the epilogue of a full block instead of 20 dependent ones (10-15 % of a launch that lives 200 us:
        // one rank's block of config 4, mode 2).  (All 20 at once -- fully unrolled -- bloated the kernel past the
        // instruction cache and spilled: slower.)
        constexpr int EPG = 4;
        // cy == 0 (the sharded engine's mode 2: T <- 0 (T 1) + A_p'u, shard_api.h c2p): what y held does not enter the result
        // and is not read -- 8 of the 18 bytes per row the epilogue loads.  (An old y that is not finite then no longer turns
        // the row into NaN; the engine's T is its own buffer, written by this product alone.)
        const bool ykeep = cy != 0.0;
        for (int rb0 = 0; rb0 < nr; rb0 += EPG * CSB_BLOCK) {     // (every wave runs every step: csb_group_max is a wave operation)
            VT yold[EPG];
            int eold[EPG];
            long long zoth[EPG];   // fused: the other splits' sums of the row (all requested before any is used)
#pragma unroll
            for (int i = 0; i < EPG; ++i) {
                const int r = rb0 + i * CSB_BLOCK + tid;
                const bool in = r < nr;
                yold[i] = (in && ykeep) ? y[row0 + r] : (VT)0;
                eold[i] = in ? (int)A.rexp[row0 + r] : 0;
                zoth[i] = 0;
            }
            if (fused) {
                // three splits' sums at a time (S = 4, 2: one round trip; all seven of S = 8 at once cost the sweep its registers)
                constexpr int ZB = 3;
                for (int k0 = 0; k0 < A.S - 1; k0 += ZB) {   // (uniform)
                    long long zv[ZB][EPG];
#pragma unroll
                    for (int kk = 0; kk < ZB; ++kk) {
                        const int k = k0 + kk;
                        const int so = k < sp ? k : k + 1;   // the k-th split that is not this one
#pragma unroll
                        for (int i = 0; i < EPG; ++i) {
                            const int r = rb0 + i * CSB_BLOCK + tid;
                            zv[kk][i] = (k < A.S - 1 && r < nr) ? A.z[(size_t)so * A.rows + row0 + r] : 0;
                        }
                    }
#pragma unroll
                    for (int kk = 0; kk < ZB; ++kk)
#pragma unroll
                        for (int i = 0; i < EPG; ++i) zoth[i] += zv[kk][i];
                }
            }
#pragma unroll
            for (int i = 0; i < EPG; ++i) {
                const int rb = rb0 + i * CSB_BLOCK;
                if (rb < nr) {                   // (uniform)
                    const int r = rb + tid;
                    unsigned hv = 0u;
                    if (r < nr) {
                        const double sum = csb_row_sum(A, (long long)acc[r] + zoth[i], row0 + r, gr, big_b, eold[i]);
                        acc[r] = 0ull;
                        const VT yn = (VT)(cy * ((double)yold[i] * sy) + sum);
                        y[row0 + r] = yn;
                        const double ys = (double)yn * nsc.s;
                        sq += ys * ys;
                        hv = csb_hi_up(fabs((double)yn));
                    }
                    if (pieces) csb_group_max(A.ymax, pc, gmx, inlds, g_first, (long long)row0 + rb + (tid & ~(WAVE - 1)), hv);
                }
            }
        }
        if (tid == 0) {  // the padding's dummy accumulator
            acc[A.R] = 0ull;
            s_bad = 0;
            s_big = 0;
        }
        if (bad_b) {  // uniform
            __syncthreads();
            if (fused)   // every chunk of the block, whichever split swept it
                csb_add_outliers<VT, NARROW>(A, aval, A.cptr[b], A.cptr[b + 1], x, sx, tau, pmax2, reinterpret_cast<double *>(acc),
                                             y, row0, nr);
            else
                for (int ri = 0; ri < nranges; ++ri) {
                    long long c0, c1;
                    range(ri, c0, c1);
                    csb_add_outliers<VT, NARROW>(A, aval, c0, c1, x, sx, tau, pmax2, reinterpret_cast<double *>(acc), y, row0, nr);
                }
            sq = csb_sumsq_rows<VT>(y, row0, nr, nsc);
            if (A.ymax != nullptr) csb_group_max_rows<VT>(A.ymax, pc, gmx, inlds, g_first, y, row0, nr);
        }
        sq = wave_sum(sq);
        if (lane == 0) red[w] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < CSB_WAVES; ++i) t += red[i];
            partials[b] = t;
        }
        if (A.ymax != nullptr) csb_group_flush(A.ymax, pc, gmx, inlds, g_first, ng);
        __syncthreads();
        if (pb && u == wg) pb[5] = wall_clock64();
    }
}

// The second launch of a column-split product: y and the partials of sum (y ns)^2 from the splits' exact sums.
// Q workgroups per row block, each with a Q-th of the block's rows and a partial of its own (partials[b * Q + qi]):
// one rank's block of config 4 at N = 8 has 64 row blocks -- with one workgroup each the launch kept a quarter of
// the chip busy and took 43 us for 100 MB.  y is what the unsplit kernel would have produced, bit for bit.
template <typename VT = double, bool NARROW = false>
__global__ __launch_bounds__(CSB_BLOCK) void k_csb_combine(
    CsbMat A, const VT *__restrict__ x, VT *__restrict__ y, const SpmvCoef *__restrict__ coef,
    const int *__restrict__ stop, double *__restrict__ partials, const double *__restrict__ pin, int npin,
    const NormSlot *__restrict__ slot_in, int skip_if_zero, CsbX xb, NScale nsc)
{
    __shared__ double accd[CSB_NACC];   // the outlier pass only (all zero otherwise)
    __shared__ double red[CSB_WAVES + 2];
    __shared__ int hist[CSB_XHIST];
    __shared__ unsigned gmx[CSB_GMX];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Q = A.Q;
    const CsbPieces pc = csb_pieces(A.rows);
    if (tid < CSB_GMX) gmx[tid] = 0u;   // (the barriers of csb_coef / csb_grids come before its first use)
    if (*stop != 0) {
        // The solve stopped while this product was under way (the scalar rider travels with its first sweep): some
        // sweeps may have run and added to zc, y is no longer wanted.  What they left behind must still go -- the
        // next product of this matrix expects zc all zero and the flags down.
        for (int u = blockIdx.x; u < A.nrb * Q; u += gridDim.x) {
            const int b = u / Q, qi = u % Q;
            const int flags = A.bad[b * CSB_QMAX + qi];
            __syncthreads();
            if (flags & 2) {
                const int row0 = A.rstart[b], nr = A.rstart[b + 1] - row0;
                const int rlo = (int)((long long)nr * qi / Q), rhi = (int)((long long)nr * (qi + 1) / Q);
                for (int r = rlo + tid; r < rhi; r += CSB_BLOCK)
                    __hip_atomic_store((unsigned long long *)&A.zc[row0 + r], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (flags != 0 && tid == 0) A.bad[b * CSB_QMAX + qi] = 0;
        }
        return;
    }
    // the coefficients and the grids: handed over by the sweeps (CsbHand), or derived like theirs -- same partials, same functions
    double sx, sy, cy;
    CsbGrid gr;
    if (A.hand != nullptr) {
        const CsbHand hd = *A.hand;
        if (hd.skip) return;
        sx = hd.sx; sy = hd.sy; cy = hd.cy;
        gr.ef = hd.ef; gr.ec = hd.ec;
        __syncthreads();   // (gmx is zero before its first use)
    } else {
        const CsbCoef co = csb_coef(coef, pin, npin, slot_in, skip_if_zero, nsc, red);
        if (co.skip) return;
        sx = co.sx; sy = co.sy; cy = co.cy;
        gr = csb_grids(xb, sx, red, hist);
    }
    const VT *__restrict__ aval = static_cast<const VT *>(A.val);
    bool cleared = false;
    for (int u = blockIdx.x; u < A.nrb * Q; u += gridDim.x) {
        const int b = u / Q, qi = u % Q;
        const int row0 = A.rstart[b];
        const int nr = A.rstart[b + 1] - row0;
        const int rlo = (int)((long long)nr * qi / Q), rhi = (int)((long long)nr * (qi + 1) / Q);
        const int flags = A.bad[b * CSB_QMAX + qi];   // uniform: what the splits of this block ran into (k_spmv_csb)
        __syncthreads();              // (everyone has read the word: thread 0 may clear it below)
        double sq = 0.0;
        const bool pieces = A.ymax != nullptr && !(flags & 1);
        const long long g_first = ((long long)row0 + rlo) >> pc.L;
        const int ng = rhi > rlo ? (int)((((long long)row0 + rhi - 1) >> pc.L) - g_first) + 1 : 0;
        const bool inlds = ng <= CSB_GMX;
        // THREE steps' worth of loads -- the splits' sums, y, the rows' exponents -- are requested before any of them is used
        // (round 6; like the sweep's epilogue since round 5): a workgroup's share of a full block (5088 rows at Q = 4) is
        // TWO round trips to memory instead of five dependent ones of a launch that lives 22 us.
        constexpr int CG = 3, ZB4 = 4;   // (five steps at once spilled 24 registers: three, i.e. two round trips for a share of 5088 rows)
        const bool ykeep = cy != 0.0;   // (cy == 0: what y held does not enter the result and is not read, csb.h k_spmv_csb)
        for (int rb0 = rlo; rb0 < rhi; rb0 += CG * CSB_BLOCK) {     // (every wave runs every step: csb_group_max is a wave operation)
            VT yo[CG];
            int eo[CG];
            long long zs[CG];
#pragma unroll
            for (int i = 0; i < CG; ++i) {
                const int r = rb0 + i * CSB_BLOCK + tid;
                const bool in = r < rhi;
                yo[i] = (in && ykeep) ? y[row0 + r] : (VT)0;
                eo[i] = in ? (int)A.rexp[row0 + r] : 0;
                zs[i] = 0;
            }
            for (int k0 = 0; k0 < A.S; k0 += ZB4) {   // (uniform; S <= 4: once)
                long long zv[ZB4][CG];
#pragma unroll
                for (int kk = 0; kk < ZB4; ++kk)
#pragma unroll
                    for (int i = 0; i < CG; ++i) {
                        const int r = rb0 + i * CSB_BLOCK + tid;
                        zv[kk][i] = (k0 + kk < A.S && r < rhi)
                                        ? __builtin_nontemporal_load(&A.z[(size_t)(k0 + kk) * A.rows + row0 + r]) : 0;
                    }
#pragma unroll
                for (int kk = 0; kk < ZB4; ++kk)
#pragma unroll
                    for (int i = 0; i < CG; ++i) zs[i] += zv[kk][i];
            }
#pragma unroll
            for (int i = 0; i < CG; ++i) {
                const int rb = rb0 + i * CSB_BLOCK;
                if (rb < rhi) {   // (uniform)
                    const int r = rb + tid;
                    unsigned hv = 0u;
                    if (r < rhi) {
                        const double sum = csb_row_sum(A, zs[i], row0 + r, gr, (flags & 2) != 0, eo[i]);
                        const VT yn = (VT)(cy * ((double)yo[i] * sy) + sum);
                        y[row0 + r] = yn;
                        const double ys = (double)yn * nsc.s;
                        sq += ys * ys;
                        hv = csb_hi_up(fabs((double)yn));
                    }
                    if (pieces) csb_group_max(A.ymax, pc, gmx, inlds, g_first, (long long)row0 + rb + (tid & ~(WAVE - 1)), hv);
                }
            }
        }
        if (flags & 1) {  // a split of this block left products out
            if (!cleared) {
                for (int i = tid; i < CSB_NACC; i += CSB_BLOCK) accd[i] = 0.0;
                cleared = true;
            }
            __syncthreads();   // (y of this share is written; accd is zero)
            csb_add_outliers<VT, NARROW>(A, aval, A.cptr[b], A.cptr[b + 1], x, sx, ldexp(1.0, gr.ef), ldexp(1.0, gr.ec), accd, y,
                                         row0, rhi, rlo);
            sq = csb_sumsq_rows<VT>(y, row0, rhi, nsc, rlo);
            if (A.ymax != nullptr) csb_group_max_rows<VT>(A.ymax, pc, gmx, inlds, g_first, y, row0, rhi, rlo);
        }
        if (flags != 0 && tid == 0) A.bad[b * CSB_QMAX + qi] = 0;
        sq = wave_sum(sq);
        if (lane == 0) red[w] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < CSB_WAVES; ++i) t += red[i];
            partials[u] = t;
        }
        if (A.ymax != nullptr) csb_group_flush(A.ymax, pc, gmx, inlds, g_first, ng);
        __syncthreads();
    }
}

}  // namespace lsqrhip
