"""Row-block sharded LSQR across the GPUs of one node (one process per GPU).

The reference is serial; this is the same iteration with A cut
into contiguous row blocks A = [A_1; ...; A_P] balanced by nonzeros:

    u, b      sharded with the rows            (m_p entries on rank p)
    v, w, x   replicated                       (n entries everywhere)

    per iteration   u_p <- A_p v - alpha u_p           local
                    |u|^2 = sum_p |u_p|^2              all-reduce, 1 double
                    T_p   = A_p' u_p                   local
                    A'u   = sum_p T_p                  all-reduce, n doubles
                    v, x, w updates + scalar recurrences: replicated, bit-identical on
                    every rank because their inputs are the all-reduced values

The collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box; "gloo" in the CPU tests).  Local work is done by a *backend*: `HipShardBackend` drives
the C-ABI stage entry points (include/lsqrhip.h, lsqrhip_shard_*); the CPU tests inject a
numpy backend with the same interface to exercise this driver under gloo.  There is no CPU
fallback in the product: constructing `HipShardBackend` without a device raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

# stage ids == csrc/shard_api.h
ST_SUMSQ_B, ST_INIT_BETA_ATU, ST_INIT_V, ST_MODE1, ST_S1_ATU, ST_VCOMBINE_UPDATE = range(6)


def partition_rows(m: int, nparts: int, weights: np.ndarray | None = None) -> list[tuple[int, int]]:
    """Contiguous row blocks [(row0, nrows)] * nparts, balanced by `weights` (nonzeros per
    row; uniform when None).  Every block is non-empty when m >= nparts."""
    if nparts < 1:
        raise ValueError("nparts must be >= 1")
    if weights is None:
        cuts = [(m * p) // nparts for p in range(nparts + 1)]
    else:
        w = np.asarray(weights, dtype=np.float64)
        if w.shape != (m,):
            raise ValueError("weights must have one entry per row")
        c = np.concatenate([[0.0], np.cumsum(w + 1.0)])       # +1: a row costs work even if empty
        targets = c[-1] * np.arange(1, nparts) / nparts
        inner = np.searchsorted(c, targets, side="left")
        cuts = [0] + [int(t) for t in inner] + [m]
        if m >= nparts:                                       # no empty block
            for p in range(1, nparts):
                cuts[p] = min(max(cuts[p], cuts[p - 1] + 1), m - (nparts - p))
        else:
            for p in range(1, nparts):
                cuts[p] = min(max(cuts[p], cuts[p - 1]), m)
    return [(cuts[p], cuts[p + 1] - cuts[p]) for p in range(nparts)]


def local_block(irow, icol, a, b, row0: int, nrows: int):
    """Rows [row0, row0+nrows) of a global 1-based COO system, renumbered from 1 (order kept)."""
    irow = np.asarray(irow)
    sel = (irow > row0) & (irow <= row0 + nrows)
    return ((irow[sel] - row0).astype(np.int32), np.asarray(icol)[sel].astype(np.int32),
            np.asarray(a)[sel].astype(np.float64), np.asarray(b)[row0:row0 + nrows].astype(np.float64))


@dataclass
class ShardResult:
    x: object            # backend-specific handle to the replicated solution (tensor / ndarray)
    istop: int
    itn: int
    anorm: float
    acond: float
    rnorm: float
    arnorm: float
    xnorm: float
    se: object = None


class TorchComm:
    """Sum all-reduce of float64 tensors + an agreement check, over torch.distributed."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce_sum(self, t):
        if getattr(t, "is_cuda", False) and self.dist.get_backend(self.group) != "nccl":
            # gloo (tests): stage through host memory on the CURRENT stream, explicitly
            tmp = t.cpu()
            self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(tmp)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def agree_max(self, value: int) -> int:
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64)
        if self.dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def barrier(self):
        self.dist.barrier(group=self.group)


class HipShardBackend:
    """Local stages on the GPU through the C-ABI (lsqrhip_shard_*).  `solver` holds A_p."""

    def __init__(self, solver, m_global: int):
        import torch
        from . import capi
        if not torch.cuda.is_available() or capi.device_count() < 1:
            raise capi.LsqrHipError(capi.ERR_NO_DEVICE, "no usable gfx950 (MI355X) device; the HIP path has no CPU fallback")
        solver._need()
        self.capi = capi
        self.torch = torch
        self.solver = solver
        self.h = solver._h
        self.m_global = int(m_global)
        self.n = solver.n
        # all library work and all collectives are ordered on ONE side stream: RCCL's internal
        # stream waits on it before a collective and it waits on the collective afterwards
        self.stream = torch.cuda.Stream()
        capi.check(capi.lib().lsqrhip_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)))
        with torch.cuda.stream(self.stream):
            self.T = torch.zeros(max(self.n, 1), dtype=torch.float64, device="cuda")
            self.sums = torch.zeros(2, dtype=torch.float64, device="cuda")
            self.x = torch.zeros(max(self.n, 1), dtype=torch.float64, device="cuda")
            self.se = None
        self.wantse = False

    def run(self, fn):
        with self.torch.cuda.stream(self.stream):
            return fn()

    def begin(self, d_b_local: int, damp, atol, btol, conlim, itnlim, wantse):
        self.wantse = bool(wantse)
        if wantse and self.se is None:
            with self.torch.cuda.stream(self.stream):
                self.se = self.torch.zeros(max(self.n, 1), dtype=self.torch.float64, device="cuda")
        self.capi.check(self.capi.lib().lsqrhip_shard_begin(
            self.h, d_b_local, self.m_global, float(damp), float(atol), float(btol), float(conlim), int(itnlim),
            int(bool(wantse)), self.T.data_ptr(), self.sums.data_ptr()))

    def stage(self, k: int):
        self.capi.check(self.capi.lib().lsqrhip_shard_stage(self.h, int(k)))

    def poll(self):
        out = (C.c_int * 3)()
        self.capi.check(self.capi.lib().lsqrhip_shard_poll(self.h, out))
        return out[0], out[1], out[2]

    def end(self) -> ShardResult:
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        self.capi.check(self.capi.lib().lsqrhip_shard_end(
            self.h, self.x.data_ptr(), self.se.data_ptr() if self.wantse else None, C.addressof(istop),
            C.addressof(itn), *[C.addressof(s) for s in sc]))
        return ShardResult(self.x[:self.n], istop.value, itn.value, *[s.value for s in sc],
                           se=self.se[:self.n] if self.wantse else None)

    def close(self):
        self.capi.check(self.capi.lib().lsqrhip_set_stream(self.h, None))


class ShardedLSQR:
    """The iteration driver: stages of a backend interleaved with the two collectives."""

    def __init__(self, backend, comm, poll_every: int = 8):
        self.be = backend
        self.comm = comm
        self.poll_every = max(1, int(poll_every))

    def _ar_scalar(self):
        self.comm.all_reduce_sum(self.be.sums[:1])

    def _ar_vector(self):
        self.comm.all_reduce_sum(self.be.T)

    def solve(self, b_local, damp=0.0, atol=0.0, btol=0.0, conlim=0.0, itnlim=100, wantse=False) -> ShardResult:
        be = self.be

        def body():
            be.begin(b_local, damp, atol, btol, conlim, itnlim, wantse)
            be.stage(ST_SUMSQ_B)
            self._ar_scalar()
            be.stage(ST_INIT_BETA_ATU)
            self._ar_vector()
            be.stage(ST_INIT_V)
            stop, itn, _ = be.poll()
            stop = self.comm.agree_max(stop)
            launched = 0
            while not stop:
                # never enqueue past itnlim: the state machine stops itself there (istop = 5)
                batch = min(self.poll_every, max(1, itnlim - launched))
                for _ in range(batch):
                    be.stage(ST_MODE1)
                    self._ar_scalar()
                    be.stage(ST_S1_ATU)
                    self._ar_vector()
                    be.stage(ST_VCOMBINE_UPDATE)
                launched += batch
                stop, itn, _ = be.poll()
                # every rank computes the same scalars from the same all-reduced inputs; the
                # MAX makes a disagreement (a bug) end the loop everywhere instead of hanging
                stop = self.comm.agree_max(stop)
            return be.end()

        run = getattr(be, "run", None)
        return run(body) if run else body()
