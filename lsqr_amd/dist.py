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
    """The two exchanges of the sharded iteration + an agreement check, over torch.distributed.

    The n-vector sum has two shapes:
      * "direct" (default): reduce-scatter as an all-to-all of n/P slices (all 7 xGMI links of a
        GPU carry one slice each, at once), a local sum of the P received slices in RANK ORDER
        (`sum_chunks`: deterministic, every element is summed once by its owner), then an
        all-gather of the reduced slices.  xGMI is point-to-point: a ring all-reduce of the same
        8n bytes is per-link bound.
      * "ring": one `all_reduce` (RCCL's choice of algorithm) -- LSQR_DIST_ALLREDUCE=ring.
    With gloo (CPU tests, no all-to-all) the direct shape is emulated by an all-gather of the
    whole vector and the same rank-ordered sum, so results are bit-identical to the RCCL path.
    """

    def __init__(self, group=None, vector_sum: str | None = None):
        import os
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.vector_sum = vector_sum or os.environ.get("LSQR_DIST_ALLREDUCE", "direct")
        self._scratch = {}

    def all_reduce_sum(self, t):
        if getattr(t, "is_cuda", False) and self.dist.get_backend(self.group) != "nccl":
            # gloo (tests): stage through host memory on the CURRENT stream, explicitly
            tmp = t.cpu()
            self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(tmp)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def _buffers(self, t, chunk):
        key = (t.device, chunk)
        if key not in self._scratch:
            import torch
            P = self.world
            self._scratch[key] = (torch.zeros(P * chunk, dtype=t.dtype, device=t.device),   # send (padded T)
                                  torch.zeros(P * chunk, dtype=t.dtype, device=t.device),   # recv (P slices)
                                  torch.zeros(chunk, dtype=t.dtype, device=t.device))       # my reduced slice
        return self._scratch[key]

    def all_reduce_vector(self, t, sum_chunks=None):
        """t <- sum over ranks of t (n doubles).  `sum_chunks(recv, P, chunk, out)` is the backend's
        rank-ordered local sum (HIP kernel on the GPU path); None = do it with torch ops."""
        import torch
        import os
        P = self.world
        forced = os.environ.get("LSQR_DIST_ALLREDUCE") == "direct!"   # exercise the direct path even at P = 1
        if (P == 1 and not forced) or self.vector_sum == "ring":
            return self.all_reduce_sum(t)
        n = t.numel()
        chunk = (n + P - 1) // P
        if self.backend == "nccl":
            send, recv, mine = self._buffers(t, chunk)
            exact = n == P * chunk                 # no padding needed: exchange t in place
            src = t if exact else send
            if not exact:
                send[:n].copy_(t)
            self.dist.all_to_all_single(recv, src, group=self.group)        # slice j of every rank -> rank j
            if sum_chunks is not None:
                sum_chunks(recv, P, chunk, mine)
            else:
                mine.copy_(recv[:chunk])
                for r in range(1, P):
                    mine.add_(recv[r * chunk:(r + 1) * chunk])
            self.dist.all_gather_into_tensor(src, mine, group=self.group)    # reduced slices -> everyone
            if not exact:
                t.copy_(send[:n])
            return
        # gloo (tests): same arithmetic -- every element summed in rank order -- via an all-gather
        src = t.cpu() if getattr(t, "is_cuda", False) else t
        parts = [torch.empty_like(src) for _ in range(P)]
        self.dist.all_gather(parts, src.contiguous(), group=self.group)
        acc = parts[0].clone()
        for r in range(1, P):
            acc.add_(parts[r])
        t.copy_(acc)

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
            self.sums = torch.zeros(4, dtype=torch.float64, device="cuda")
            self.x = torch.zeros(max(self.n, 1), dtype=torch.float64, device="cuda")
            self.se = None
        self.wantse = False
        self.nsums_b = 3          # stage 0 leaves Blue's three sums of b^2 in sums[0..2]

    def agree_norm_scale(self, comm):
        e = C.c_int64()
        self.capi.check(self.capi.lib().lsqrhip_get_option(self.h, b"norm_exp", C.byref(e)))
        e_all = comm.agree_max(int(e.value) + 4096) - 4096
        self.capi.check(self.capi.lib().lsqrhip_set_option(self.h, b"norm_exp", int(e_all)))

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

    def sum_chunks(self, recv, nchunks: int, chunk: int, out):
        """Rank-ordered sum of the slices received in the direct reduce-scatter (HIP kernel)."""
        self.capi.check(self.capi.lib().lsqrhip_sum_chunks(self.h, recv.data_ptr(), int(nchunks), int(chunk),
                                                           out.data_ptr()))

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

    def _ar_scalar(self, k: int = 1):
        self.comm.all_reduce_sum(self.be.sums[:k])

    def _ar_vector(self):
        arv = getattr(self.comm, "all_reduce_vector", None)
        if arv is None:
            self.comm.all_reduce_sum(self.be.T)
        else:
            arv(self.be.T, getattr(self.be, "sum_chunks", None))

    def solve(self, b_local, damp=0.0, atol=0.0, btol=0.0, conlim=0.0, itnlim=100, wantse=False) -> ShardResult:
        be = self.be

        def body():
            # the ranks scale their fused sums of squares by ONE power of two (csrc/scalar.h
            # "range-safe norms"): the largest of the exponents their row blocks call for
            agree = getattr(be, "agree_norm_scale", None)
            if agree is not None:
                agree(self.comm)
            be.begin(b_local, damp, atol, btol, conlim, itnlim, wantse)
            be.stage(ST_SUMSQ_B)
            # norm(b): three range-safe partial sums (small / mid / big elements), additive over ranks
            self._ar_scalar(getattr(be, "nsums_b", 1))
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
