// state.h -- device-resident scalar state of one LSQR solve.
//
// All ~35 scalars of the reference's iteration (src/lsqr.f90:565-574) live in
// one struct in HBM.  The vector kernels read their coefficients from it and
// the one-workgroup scalar kernels advance it, so an iteration needs no host
// round trip; the host only polls `stop` between captured batches.
#pragma once

#include <stdint.h>

namespace lsqrhip {

// y_new = cy * (y * sy) + sum_j A_ij * (x_j * sx)      (see spmv.h)
struct SpmvCoef {
    double sx, sy, cy;
    int skip;  // launch is a no-op (mode 2 when beta == 0, src/lsqr.f90:691)
    int pad;
};

// What one SpMV hands to the next in the pipelined schedule (spmv.h, solve_loop.h): the
// norm it derived from the previous kernel's partials and the matching scale.
struct NormSlot {
    double nrm;    // beta (published by mode 2) or alpha (published by mode 1)
    double scale;  // 1/nrm, or 1 when nrm == 0  (the guards at src/lsqr.f90:635, 641, 691, 696)
};

// Power-of-two scale inside the fused sums of squares: partial += (y * s)^2, norm = sqrt(sum) * inv.
struct NScale {
    double s, inv;
};

struct LsqrState {
    // control ------------------------------------------------------------
    int stop;      // != 0: every kernel of the loop returns at once
    int istop;     // src/lsqr.f90:520-538
    int itn;
    int nstop;
    int itnlim;
    int maxdx;
    int damped;
    int wantse;
    int want_log;
    int log_cap;    // records the log buffer can hold
    int log_count;  // records written so far (only iterations the reference would print)
    int log_truncated;  // records had to be dropped (the buffer keeps its last slot for the final iteration)
    int m, n;
    // user tolerances ------------------------------------------------------
    double damp, atol, btol, ctol;
    // Golub-Kahan scalars ----------------------------------------------------
    double alpha, beta;
    void *xout;     // device address x goes to when the stop flag rises (vec.h k_out_copy), or null
    int batch;      // batches settled so far in this solve (k_s3 with a snapshot: which of the two host slots is next)
    int wp32;       // REAL32 handle: the "1 + test <= 1" stops are taken in real32 like the reference's REAL32 build
    double ns_inv;  // fused norms are sqrt(sum of (y * ns)^2) * ns_inv, ns a power of two (scalar.h "range-safe norms")
    double su;  // pending scale of U: u = U * su   (1/beta, or 1 when beta == 0)
    double sv;  // pending scale of V: v = V * sv   (1/alpha, or 1 when alpha == 0)
    SpmvCoef c1;    // mode-1 launch of the next iteration: U <- (-alpha)*(U*su) + A (V*sv)
    SpmvCoef c2;    // mode-2 launch of this iteration:     V <- (-beta)*(V*sv) + A'(U*su)
    SpmvCoef c2p;   // row-sharded mode 2: T_p <- A_p'(U_p*su) (then V <- c2.cy*(V*c2.sy) + sum_p T_p)
    // rotations / estimates (names as in the reference) ------------------------
    // rhobar / phibar are kept by iteration parity: step 2 of iteration k reads [(k-1)&1] and
    // writes [k&1] ([0] = the initial values).  The fused x/w update of iteration k runs
    // inside the mode-1 kernel of iteration k+1 and recomputes the rotation from [(k-1)&1]
    // while that kernel's rider writes [k&1] (solve_loop.h "fused update").
    double rhobar2[2], phibar2[2];
    double anorm, acond, dnorm, dxmax, res2, psi;
    double xnorm, xnorm1, cs2, sn2, z, bnorm, rnorm, arnorm;
    double rho, phi, theta, tau;  // S2 -> S3
    double t1, t2, t3;            // coefficients of the x/w update kernel
    // log-only extras ----------------------------------------------------------
    double alpha0, beta0, test2_0;
    // The seal of a host snapshot (scalar.h k_s3_snap): 0 in the device state; in a pinned host slot the kernel writes
    // batch + 1 LAST, behind a system-scope fence -- a host that reads it holds every other word of the snapshot.
    int seal;
    int pad_seal;
};

constexpr int LOG_STRIDE = 14;  // == LSQRHIP_LOG_STRIDE

}  // namespace lsqrhip
