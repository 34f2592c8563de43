"""A numpy stand-in for the per-rank stage kernels (csrc/shard_api.h) -- TESTS ONLY.

Lets the CPU suite drive lsqr_amd.dist.ShardedLSQR (partitioner, stage order, the two
collectives, stop agreement) under the gloo backend with world_size > 1.  Local products use
the oracle's aprod; the scalar steps restate csrc/scalar.h.  Nothing in the product imports this.
"""
import math

import numpy as np
import torch

import oracle
from lsqr_amd.dist import (ST_INIT_BETA_ATU, ST_INIT_V, ST_MODE1, ST_S1_ATU, ST_SUMSQ_B,
                           ST_VCOMBINE_UPDATE, ShardResult)


def d2norm(a, b):
    s = abs(a) + abs(b)
    return 0.0 if s == 0.0 else s * math.sqrt((a / s) ** 2 + (b / s) ** 2)


class NumpyShardBackend:
    def __init__(self, m_local, n, irow, icol, a, m_global):
        self.mp, self.n, self.mg = m_local, n, m_global
        self.coo = (np.ascontiguousarray(irow, np.int32), np.ascontiguousarray(icol, np.int32),
                    np.ascontiguousarray(a, np.float64))
        self._T = np.zeros(max(n, 1))
        self._sums = np.zeros(2)
        self.T = torch.from_numpy(self._T)        # shared memory: the driver all-reduces these
        self.sums = torch.from_numpy(self._sums)
        self.po = oracle.port()

    def _ax(self, v):     # A_p v
        return self.po.aprod(1, self.mp, self.n, *self.coo, v, np.zeros(self.mp))[1]

    def _atu(self, u):    # A_p' u
        return self.po.aprod(2, self.mp, self.n, *self.coo, np.zeros(self.n), u)[0]

    def begin(self, b_local, damp, atol, btol, conlim, itnlim, wantse):
        s = self.s = dict(damp=damp, atol=atol, btol=btol, ctol=1.0 / conlim if conlim > 0 else 0.0,
                          itnlim=itnlim, wantse=wantse, damped=damp > 0, stop=0, istop=0, itn=0, nstop=0,
                          anorm=0.0, acond=0.0, dnorm=0.0, res2=0.0, psi=0.0, xnorm=0.0, xnorm1=0.0,
                          cs2=-1.0, sn2=0.0, z=0.0, alpha=0.0, beta=0.0, skip=True, rnorm=0.0, arnorm=0.0,
                          bnorm=0.0)
        self.u = np.array(b_local, dtype=np.float64)
        self.v = np.zeros(self.n)
        self.w = np.zeros(self.n)
        self.x = np.zeros(self.n)
        self.se = np.zeros(self.n)
        self._T[:] = 0.0
        self._sums[:] = 0.0

    def stage(self, k):
        s = self.s
        if k == ST_SUMSQ_B:
            self._sums[0] = float(np.dot(self.u, self.u))
        elif k == ST_INIT_BETA_ATU:
            s["beta"] = math.sqrt(self._sums[0])
            s["skip"] = not (s["beta"] > 0)
            if not s["skip"]:
                self.u *= 1.0 / s["beta"]
                self._T[:self.n] = self._atu(self.u)
        elif k == ST_INIT_V:
            if not s["skip"]:
                self.v = self._T[:self.n].copy()
                s["alpha"] = math.sqrt(float(np.dot(self.v, self.v)))
            if s["alpha"] > 0:
                self.v *= 1.0 / s["alpha"]
                self.w = self.v.copy()
            s["arnorm"] = s["alpha"] * s["beta"]
            s["bnorm"] = s["rnorm"] = s["beta"]
            if s["arnorm"] == 0.0:
                s["stop"] = 1
            s["rhobar"], s["phibar"] = s["alpha"], s["beta"]
        elif k == ST_MODE1:
            if s["stop"]:
                return
            self.u = -s["alpha"] * self.u + self._ax(self.v)
            self._sums[0] = float(np.dot(self.u, self.u))
        elif k == ST_S1_ATU:
            if s["stop"]:
                return
            s["itn"] += 1
            s["beta"] = math.sqrt(self._sums[0])
            s["anorm"] = d2norm(s["anorm"], d2norm(d2norm(s["alpha"], s["beta"]), s["damp"]))
            s["skip"] = not (s["beta"] > 0)
            if not s["skip"]:
                self.u *= 1.0 / s["beta"]
                self._T[:self.n] = self._atu(self.u)
        elif k == ST_VCOMBINE_UPDATE:
            if s["stop"]:
                return
            if not s["skip"]:
                self.v = -s["beta"] * self.v + self._T[:self.n]
                s["alpha"] = math.sqrt(float(np.dot(self.v, self.v)))
                if s["alpha"] > 0:
                    self.v *= 1.0 / s["alpha"]
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
            d = (t3 * self.w) ** 2
            if s["wantse"]:
                self.se += d
            self.x += t1 * self.w
            self.w = t2 * self.w + self.v
            dknorm = math.sqrt(float(np.sum(d)))
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

    def end(self):
        s = self.s
        se = None
        if s["wantse"]:
            t = 1.0
            if self.mg > self.n: t = float(self.mg - self.n)
            if s["damped"]: t = float(self.mg)
            se = (s["rnorm"] / math.sqrt(t)) * np.sqrt(self.se) if s["itn"] > 0 else self.se
        istop = 3 if (s["damped"] and s["istop"] == 2) else s["istop"]
        return ShardResult(self.x.copy(), istop, s["itn"], s["anorm"], s["acond"], s["rnorm"], s["arnorm"],
                           s["xnorm"], se=se)
