"""Row-block sharded LSQR across the GPUs of one node (one process per GPU).

The reference is serial; this is the same iteration with A cut
into contiguous row blocks A = [A_1; ...; A_P] balanced by nonzeros:

    u, b      sharded with the rows                          (m_p entries on rank p)
    v         replicated (mode 1 gathers from all of it)     (P * chunk entries, chunk = ceil(n / P))
    x, w, se  sharded by column slices                       (rank q owns [q chunk, (q + 1) chunk))

    per iteration   u_p <- A_p v - alpha u_p                       local
                    |u|^2 = sum_p |u_p|^2                          all-reduce, 1 double
                    T_p   = A_p' u_p                               local
                    slice q of every T_p -> rank q, summed there   direct reduce-scatter (rank order)
                    v_q <- T_q - beta v_q ; |v_q|^2 , |w_q|^2      all-reduce, 2 doubles
                    rotations, x_q, w_q, stopping tests            slice-local + replicated scalars
                    all-gather of the v slices

Two drivers over the same stages (csrc/shard_api.h):
  * `EngineSolver`: the whole loop runs in C++ (csrc/shard_engine.h: kernels AND RCCL calls are
    enqueued by the library, lsqrhip_shard_solve); Python only hands the RCCL id around.  This is
    what bench.py --gpus N uses.
  * `ShardedLSQR`: the loop in Python, stage by stage, with the exchanges through
    torch.distributed.  Local work is done by a *backend*: `HipShardBackend` drives the C-ABI stage
    entry points; the CPU tests inject a numpy backend with the same interface to exercise this
    driver under gloo.  Both drivers sum the ranks' scalars and vector slices in RANK ORDER, so
    the CPU tests pin the arithmetic of the RCCL path bit for bit.
There is no CPU fallback in the product: constructing `HipShardBackend` / `EngineSolver` without a
device raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

# stage ids == csrc/shard_api.h
(ST_SUMSQ_B, ST_INIT_BETA_ATU, ST_INIT_V, ST_INIT_W, ST_MODE1, ST_S1_ATU, ST_VCOMBINE, ST_UPDATE) = range(8)


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
    """The exchanges of the sharded iteration over torch.distributed ("nccl" = RCCL, or gloo).

    Every reduction is an exchange of the ranks' contributions followed by a LOCAL sum in rank
    order -- scalars: all-gather of the 1-3 doubles; the n-vector: all-to-all of the column
    slices (the direct reduce-scatter: all 7 xGMI links of a GPU carry one slice each, at once; a
    ring all-reduce of the same 8n bytes is per-link bound) -- so that the
    result does not depend on the collective's algorithm and every rank holds identical bits.
    """

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def _host(self, t):
        return t.cpu() if getattr(t, "is_cuda", False) else t

    def all_reduce_scalars(self, t):
        """t (k doubles) <- sum over ranks, added in rank order."""
        import torch
        if self.world == 1:
            return
        src = t if self.backend == "nccl" else self._host(t)
        parts = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(parts, src.contiguous(), group=self.group)
        acc = parts[0].clone()
        for r in range(1, self.world):
            acc.add_(parts[r])
        t.copy_(acc)

    def scatter_slices(self, T, R, chunk: int):
        """R[r * chunk : (r+1) * chunk] <- rank r's T[rank * chunk : (rank+1) * chunk]."""
        import torch
        P = self.world
        if P == 1:
            R[:chunk].copy_(T[:chunk])
            return
        if self.backend == "nccl":
            self.dist.all_to_all_single(R, T, group=self.group)
            return
        src = self._host(T)                      # gloo (tests): an all-gather of everything, then pick
        parts = [torch.empty_like(src) for _ in range(P)]
        self.dist.all_gather(parts, src.contiguous(), group=self.group)
        got = torch.cat([parts[r][self.rank * chunk:(self.rank + 1) * chunk] for r in range(P)])
        R.copy_(got)

    def gather_slices(self, V, chunk: int):
        """Slice q of V <- rank q's slice q (in place)."""
        import torch
        P = self.world
        if P == 1:
            return
        mine = V[self.rank * chunk:(self.rank + 1) * chunk]
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(V, mine.clone(), group=self.group)
            return
        src = self._host(mine).contiguous()
        parts = [torch.empty_like(src) for _ in range(P)]
        self.dist.all_gather(parts, src, group=self.group)
        V.copy_(torch.cat(parts))

    def agree_max(self, value: int) -> int:
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64)
        if self.backend == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def barrier(self):
        self.dist.barrier(group=self.group)


class HipShardBackend:
    """Local stages on the GPU through the C-ABI (lsqrhip_shard_*).  `solver` holds A_p."""

    def __init__(self, solver, m_global: int, world: int, rank: int):
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
        self.world, self.rank = int(world), int(rank)
        self.chunk = (self.n + self.world - 1) // self.world
        # all library work and all collectives are ordered on ONE side stream: RCCL's internal
        # stream waits on it before a collective and it waits on the collective afterwards
        self.stream = torch.cuda.Stream()
        capi.check(capi.lib().lsqrhip_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)))
        full = max(self.chunk * self.world, 1)
        # a REAL32 handle (src/lsqr_kinds.F90:16-17): b, the exchange buffers, x and se are float arrays -- half the
        # bytes in every collective; the scalars stay binary64
        wp = torch.float32 if getattr(solver, "real32", False) else torch.float64
        with torch.cuda.stream(self.stream):
            z = lambda k: torch.zeros(k, dtype=wp, device="cuda")     # noqa: E731
            self.T, self.R, self.V, self.x, self.se = z(full), z(full), z(full), z(full), z(full)
            self.sums = torch.zeros(4, dtype=torch.float64, device="cuda")
        self.wantse = False

    def run(self, fn):
        with self.torch.cuda.stream(self.stream):
            return fn()

    def agree_norm_scale(self, comm):
        e = C.c_int64()
        self.capi.check(self.capi.lib().lsqrhip_get_option(self.h, b"norm_exp", C.byref(e)))
        e_all = comm.agree_max(int(e.value) + 4096) - 4096
        self.capi.check(self.capi.lib().lsqrhip_set_option(self.h, b"norm_exp", int(e_all)))

    def begin(self, d_b_local: int, damp, atol, btol, conlim, itnlim, wantse):
        self.wantse = bool(wantse)
        self.capi.check(self.capi.lib().lsqrhip_shard_begin(
            self.h, d_b_local, self.m_global, self.world, self.rank, float(damp), float(atol), float(btol),
            float(conlim), int(itnlim), int(bool(wantse)), self.T.data_ptr(), self.R.data_ptr(), self.V.data_ptr(),
            self.sums.data_ptr()))

    def stage(self, k: int):
        self.capi.check(self.capi.lib().lsqrhip_shard_stage(self.h, int(k)))

    def poll(self):
        out = (C.c_int * 3)()
        self.capi.check(self.capi.lib().lsqrhip_shard_poll(self.h, out))
        return out[0], out[1], out[2]

    def end(self, comm) -> ShardResult:
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        self.capi.check(self.capi.lib().lsqrhip_shard_end(
            self.h, self.x.data_ptr(), self.se.data_ptr() if self.wantse else None, C.addressof(istop),
            C.addressof(itn), *[C.addressof(s) for s in sc]))
        comm.gather_slices(self.x, self.chunk)
        if self.wantse:
            comm.gather_slices(self.se, self.chunk)
        return ShardResult(self.x[:self.n], istop.value, itn.value, *[s.value for s in sc],
                           se=self.se[:self.n] if self.wantse else None)

    def close(self):
        self.capi.check(self.capi.lib().lsqrhip_set_stream(self.h, None))


class ShardedLSQR:
    """The iteration driver in Python: stages of a backend interleaved with the exchanges."""

    def __init__(self, backend, comm, poll_every: int = 8):
        self.be = backend
        self.comm = comm
        self.poll_every = max(1, int(poll_every))

    def solve(self, b_local, damp=0.0, atol=0.0, btol=0.0, conlim=0.0, itnlim=100, wantse=False) -> ShardResult:
        be, comm = self.be, self.comm
        chunk = be.chunk

        def body():
            # the ranks scale their fused sums of squares by ONE power of two (csrc/scalar.h
            # "range-safe norms"): the largest of the exponents their row blocks call for
            agree = getattr(be, "agree_norm_scale", None)
            if agree is not None:
                agree(comm)
            be.begin(b_local, damp, atol, btol, conlim, itnlim, wantse)
            be.stage(ST_SUMSQ_B)
            comm.all_reduce_scalars(be.sums[:3])      # norm(b): three range-safe sums, additive over ranks
            be.stage(ST_INIT_BETA_ATU)
            comm.scatter_slices(be.T, be.R, chunk)
            be.stage(ST_INIT_V)
            comm.all_reduce_scalars(be.sums[:2])
            be.stage(ST_INIT_W)
            comm.gather_slices(be.V, chunk)
            stop, itn, _ = be.poll()
            stop = comm.agree_max(stop)
            launched = 0
            while not stop:
                # never enqueue past itnlim: the state machine stops itself there (istop = 5)
                batch = min(self.poll_every, max(1, itnlim - launched))
                for _ in range(batch):
                    be.stage(ST_MODE1)
                    comm.all_reduce_scalars(be.sums[:1])
                    be.stage(ST_S1_ATU)
                    comm.scatter_slices(be.T, be.R, chunk)
                    be.stage(ST_VCOMBINE)
                    comm.all_reduce_scalars(be.sums[:2])
                    be.stage(ST_UPDATE)
                    comm.gather_slices(be.V, chunk)
                launched += batch
                stop, itn, _ = be.poll()
                # every rank computes the same scalars from the same all-reduced inputs; the
                # MAX makes a disagreement (a bug) end the loop everywhere instead of hanging
                stop = comm.agree_max(stop)
            return be.end(comm)

        run = getattr(be, "run", None)
        return run(body) if run else body()


class EngineSolver:
    """One rank of the C++ engine (csrc/shard_engine.h): lsqrhip_shard_comm_init + lsqrhip_shard_solve.
    `solver` holds A_p; the RCCL id travels through torch.distributed's object broadcast."""

    def __init__(self, solver, row0: int, m_global: int, world: int, rank: int, group=None):
        import torch
        import torch.distributed as dist
        from . import capi
        if not torch.cuda.is_available() or capi.device_count() < 1:
            raise capi.LsqrHipError(capi.ERR_NO_DEVICE, "no usable gfx950 (MI355X) device; the HIP path has no CPU fallback")
        solver._need()
        self.capi, self.torch, self.solver = capi, torch, solver
        self.n, self.world, self.rank = solver.n, int(world), int(rank)
        # Every rank takes part in every collective of this handshake whatever happened to it locally: a rank
        # that raised on its own would leave its peers blocked in the broadcast (or in the all-reduce of the
        # caller's fall-back logic).  Failures are agreed on and then raised by ALL ranks.
        ident, err = [None], None
        if world > 1:
            if rank == 0:
                try:
                    buf = C.create_string_buffer(128)
                    capi.check(capi.lib().lsqrhip_rccl_unique_id(buf))
                    ident = [bytes(buf.raw)]
                except Exception as e:  # noqa: BLE001
                    err = e
            dist.broadcast_object_list(ident, src=0, group=group)      # None: rank 0 has no id
            if ident[0] is None:
                raise err or capi.LsqrHipError(capi.ERR_HIP, "rank 0 could not obtain an RCCL unique id")
        ok = 1
        try:
            capi.check(capi.lib().lsqrhip_shard_comm_init(solver._h, self.world, self.rank, int(row0), int(m_global),
                                                          ident[0] if world > 1 else None))
        except Exception as e:  # noqa: BLE001
            ok, err = 0, e
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0:
                raise err or capi.LsqrHipError(capi.ERR_HIP, "lsqrhip_shard_comm_init failed on another rank")
        elif not ok:
            raise err
        self.d_x = capi.DeviceBuffer(8 * max(self.n, 1))
        self.d_se = None

    def solve(self, d_b_local: int, damp=0.0, atol=0.0, btol=0.0, conlim=0.0, itnlim=100, wantse=False) -> ShardResult:
        capi = self.capi
        if wantse and self.d_se is None:
            self.d_se = capi.DeviceBuffer(8 * max(self.n, 1))
        istop, itn = C.c_int(), C.c_int()
        sc = [C.c_double() for _ in range(5)]
        capi.check(capi.lib().lsqrhip_shard_solve(
            self.solver._h, d_b_local, float(damp), float(atol), float(btol), float(conlim), int(itnlim),
            int(bool(wantse)), self.d_x.ptr, self.d_se.ptr if wantse else None, C.addressof(istop), C.addressof(itn),
            *[C.addressof(s) for s in sc]))
        return ShardResult(self.d_x, istop.value, itn.value, *[s.value for s in sc], se=self.d_se if wantse else None)
