"""A numpy stand-in for the per-rank stage kernels (csrc/shard_api.h) -- TESTS ONLY.

Lets the CPU suite drive lsqr_amd.dist.ShardedLSQR (partitioner, stage order, the four exchanges,
stop agreement) under the gloo backend with world_size > 1.  Local products use the oracle's aprod;
the scalar steps restate csrc/scalar.h, the column-slice bookkeeping restates csrc/shard_api.h.
Nothing in the product imports this.
"""
import math

import numpy as np
import torch

import oracle
from lsqr_amd.dist import (ST_INIT_BETA_ATU, ST_INIT_V, ST_INIT_W, ST_MODE1, ST_S1_ATU, ST_SUMSQ_B, ST_UPDATE,
                           ST_VCOMBINE, ShardResult)


def d2norm(a, b):
    s = abs(a) + abs(b)
    return 0.0 if s == 0.0 else s * math.sqrt((a / s) ** 2 + (b / s) ** 2)


class NumpyShardBackend:
    def __init__(self, m_local, n, irow, icol, a, m_global, world, rank):
        self.mp, self.n, self.mg = m_local, n, m_global
        self.world, self.rank = world, rank
        self.chunk = (n + world - 1) // world
        self.my0 = min(rank * self.chunk, n)
        self.mylen = min(self.chunk, n - self.my0)
        self.coo = (np.ascontiguousarray(irow, np.int32), np.ascontiguousarray(icol, np.int32),
                    np.ascontiguousarray(a, np.float64))
        full = max(self.chunk * world, 1)
        self._T, self._R, self._V = np.zeros(full), np.zeros(full), np.zeros(full)
        self._x, self._se = np.zeros(full), np.zeros(full)
        self._sums = np.zeros(4)
        self.T, self.R, self.V = (torch.from_numpy(v) for v in (self._T, self._R, self._V))   # shared memory
        self.sums = torch.from_numpy(self._sums)
        self.po = oracle.port()

    def _ax(self, v):     # A_p v
        return self.po.aprod(1, self.mp, self.n, *self.coo, v, np.zeros(self.mp))[1]

    def _atu(self, u):    # A_p' u
        return self.po.aprod(2, self.mp, self.n, *self.coo, np.zeros(self.n), u)[0]

    def _slice(self, v):
        return v[self.my0:self.my0 + self.mylen]

    def _rsum(self):      # rank-ordered sum of the received slices
        t = self._R[:self.chunk].copy()
        for r in range(1, self.world):
            t = t + self._R[r * self.chunk:(r + 1) * self.chunk]
        return t[:self.mylen]

    def begin(self, b_local, damp, atol, btol, conlim, itnlim, wantse):
        self.s = dict(damp=damp, atol=atol, btol=btol, ctol=1.0 / conlim if conlim > 0 else 0.0,
                      itnlim=itnlim, wantse=wantse, damped=damp > 0, stop=0, istop=0, itn=0, nstop=0,
                      anorm=0.0, acond=0.0, dnorm=0.0, res2=0.0, psi=0.0, xnorm=0.0, xnorm1=0.0,
                      cs2=-1.0, sn2=0.0, z=0.0, alpha=0.0, beta=0.0, skip=True, rnorm=0.0, arnorm=0.0,
                      bnorm=0.0, su=1.0, sv=1.0, wsq=0.0)
        self.u = np.array(b_local, dtype=np.float64)     # U: un-normalised, with the pending scale su
        self.w = np.zeros(self.mylen)
        self.x = np.zeros(self.mylen)
        self.se = np.zeros(self.mylen)
        for v in (self._T, self._R, self._V, self._x, self._se, self._sums):
            v[:] = 0.0

    def stage(self, k):
        s = self.s
        Vq = self._slice(self._V)
        if k == ST_SUMSQ_B:
            self._sums[0], self._sums[1], self._sums[2] = 0.0, float(np.dot(self.u, self.u)), 0.0   # mid plane only
        elif k == ST_INIT_BETA_ATU:
            s["beta"] = math.sqrt(self._sums[1])
            s["skip"] = not (s["beta"] > 0)
            s["su"] = 1.0 / s["beta"] if not s["skip"] else 1.0
            if not s["skip"]:
                self._T[:self.n] = self._atu(self.u * s["su"])
        elif k == ST_INIT_V:
            if not s["skip"]:
                Vq[:] = self._rsum()
                self._sums[0] = float(np.dot(Vq, Vq))
            else:
                self._sums[0] = 0.0
            self._sums[1] = s["wsq"]
        elif k == ST_INIT_W:
            s["alpha"] = 0.0 if s["skip"] else math.sqrt(self._sums[0])
            s["sv"] = 1.0 / s["alpha"] if s["alpha"] > 0 else 1.0
            s["arnorm"] = s["alpha"] * s["beta"]
            s["bnorm"] = s["rnorm"] = s["beta"]
            if s["arnorm"] == 0.0:
                s["stop"] = 1
            else:
                self.w = Vq * s["sv"]
                s["wsq"] = float(np.dot(self.w, self.w))
            s["rhobar"], s["phibar"] = s["alpha"], s["beta"]
        elif k == ST_MODE1:
            if s["stop"]:
                return
            self.u = -s["alpha"] * (self.u * s["su"]) + self._ax(self._V[:self.n] * s["sv"])
            self._sums[0] = float(np.dot(self.u, self.u))
        elif k == ST_S1_ATU:
            if s["stop"]:
                return
            s["itn"] += 1
            s["beta"] = math.sqrt(self._sums[0])
            s["anorm"] = d2norm(s["anorm"], d2norm(d2norm(s["alpha"], s["beta"]), s["damp"]))
            s["skip"] = not (s["beta"] > 0)
            s["su"] = 1.0 / s["beta"] if not s["skip"] else 1.0
            if not s["skip"]:
                self._T[:self.n] = self._atu(self.u * s["su"])
        elif k == ST_VCOMBINE:
            if s["stop"]:
                return
            if not s["skip"]:
                Vq[:] = -s["beta"] * (Vq * s["sv"]) + self._rsum()
                self._sums[0] = float(np.dot(Vq, Vq))
            else:
                self._sums[0] = 0.0
            self._sums[1] = s["wsq"]
        elif k == ST_UPDATE:
            if s["stop"]:
                return
            if not s["skip"]:
                s["alpha"] = math.sqrt(self._sums[0])
                s["sv"] = 1.0 / s["alpha"] if s["alpha"] > 0 else 1.0
            alpha, beta, damp = s["alpha"], s["beta"], s["damp"]
            rhbar1, phibar = s["rhobar"], s["phibar"]
            if s["damped"]:
                rhbar1 = d2norm(s["rhobar"], damp)
                s["psi"] = damp / rhbar1 * phibar
                phibar = s["rhobar"] / rhbar1 * phibar
            rho = d2norm(rhbar1, beta)
            cs, sn = rhbar1 / rho, beta / rho
            theta = sn * alpha
            s["rhobar"] = -cs * alpha
            phi = cs * phibar
            s["phibar"] = sn * phibar
            tau = sn * phi
            t1, t2, t3 = phi / rho, -theta / rho, 1.0 / rho
            dknorm = math.sqrt((t3 * t3) * self._sums[1])          # |t3| sqrt(sum w^2), w = the previous iteration's
            if s["wantse"]:
                self.se += (t3 * self.w) ** 2
            self.x += t1 * self.w
            self.w = t2 * self.w + Vq * s["sv"]
            s["wsq"] = float(np.dot(self.w, self.w))
            s["dnorm"] = d2norm(s["dnorm"], dknorm)
            delta, gambar = s["sn2"] * rho, -s["cs2"] * rho
            rhs = phi - delta * s["z"]
            zbar = rhs / gambar
            s["xnorm"] = d2norm(s["xnorm1"], zbar)
            gamma = d2norm(gambar, theta)
            s["cs2"], s["sn2"], s["z"] = gambar / gamma, theta / gamma, rhs / gamma
            s["xnorm1"] = d2norm(s["xnorm1"], s["z"])
            anorm, bnorm, xnorm = s["anorm"], s["bnorm"], s["xnorm"]
            s["acond"] = anorm * s["dnorm"]
            s["res2"] = d2norm(s["res2"], s["psi"])
            rnorm = s["rnorm"] = d2norm(s["res2"], s["phibar"])
            arnorm = s["arnorm"] = alpha * abs(tau)
            test1 = rnorm / bnorm
            test2 = arnorm / (anorm * rnorm) if rnorm > 0 else 0.0
            test3 = 1.0 / s["acond"]
            t1c = test1 / (1.0 + anorm * xnorm / bnorm)
            rtol = s["btol"] + s["atol"] * anorm * xnorm / bnorm
            istop = s["istop"]
            if s["itn"] >= s["itnlim"]: istop = 5
            if 1.0 + test3 <= 1.0: istop = 4
            if 1.0 + test2 <= 1.0: istop = 2
            if 1.0 + t1c <= 1.0: istop = 1
            if test3 <= s["ctol"]: istop = 4
            if test2 <= s["atol"]: istop = 2
            if test1 <= rtol: istop = 1
            s["istop"] = istop
            if istop != 0:
                s["stop"] = 1
        else:
            raise ValueError(k)

    def poll(self):
        return self.s["stop"], self.s["itn"], self.s["istop"]

    def end(self, comm):
        s = self.s
        se_slice = self.se
        if s["wantse"]:
            t = 1.0
            if self.mg > self.n: t = float(self.mg - self.n)
            if s["damped"]: t = float(self.mg)
            se_slice = (s["rnorm"] / math.sqrt(t)) * np.sqrt(self.se) if s["itn"] > 0 else self.se
        self._x[self.my0:self.my0 + self.mylen] = self.x
        self._se[self.my0:self.my0 + self.mylen] = se_slice
        xt, st = torch.from_numpy(self._x), torch.from_numpy(self._se)
        comm.gather_slices(xt, self.chunk)
        if s["wantse"]:
            comm.gather_slices(st, self.chunk)
        istop = 3 if (s["damped"] and s["istop"] == 2) else s["istop"]
        return ShardResult(self._x[:self.n].copy(), istop, s["itn"], s["anorm"], s["acond"], s["rnorm"], s["arnorm"],
                           s["xnorm"], se=self._se[:self.n].copy() if s["wantse"] else None)
