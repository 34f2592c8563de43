"""One rank of a sharded solve (launched by tests/test_dist.py as a subprocess).

    python tests/dist_worker.py RANK WORLD PORT CASE BACKEND OUT.npz [ITNCAP]

BACKEND = numpy (CPU, gloo, oracle-backed stage stand-in) | hip / hip32 (C-ABI stages on cuda:0, gloo; binary64 / REAL32)
        | engine / engine32 / engine_ov / engine32_ov / engine_copy / engine32_copy (the C++ engine over RCCL, binary64 /
          REAL32, exchanges plain / overlapped / as copy-engine pulls over IPC-mapped buffers: the ranks SHARE cuda:0, each with a NCCL_HOSTID of its own -- lsqr_amd.dist_bench.share_one_gpu)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)     # a hang prints where and ends the rank (a loaded box: minutes, not a hang)
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    case, backend, out = sys.argv[4], sys.argv[5], sys.argv[6]
    import torch.distributed as dist
    from cases import build_cases
    from lsqr_amd.dist import ShardedLSQR, TorchComm, local_block, partition_rows

    if backend.startswith("engine"):
        return engine_rank(rank, world, port, case, backend, out)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    p, o = build_cases()[case]
    w = np.bincount(p.irow - 1, minlength=p.m).astype(np.float64)
    blocks = partition_rows(p.m, world, w)
    row0, nrows = blocks[rank]
    irow, icol, a, b = local_block(p.irow, p.icol, p.a, p.b, row0, nrows)
    comm = TorchComm()
    if backend == "numpy":
        from np_shard_backend import NumpyShardBackend
        be = NumpyShardBackend(nrows, p.n, irow, icol, a, p.m, world, rank)
        b_arg = b
    else:
        from lsqr_amd import capi
        from lsqr_amd.dist import HipShardBackend
        from lsqr_amd.solver import lsqr_solver_ez
        real32 = backend == "hip32"      # REAL32 handle: float blocks, float exchange buffers (half the bytes over gloo too)
        s = lsqr_solver_ez().initialize(nrows, p.n, a, irow, icol, real32=real32)
        be = HipShardBackend(s, p.m, world, rank)
        d_b = capi.DeviceBuffer.from_array((b if nrows else np.zeros(1)).astype(np.float32 if real32 else np.float64))
        b_arg = d_b.ptr.value
    r = ShardedLSQR(be, comm, poll_every=3).solve(b_arg, damp=o["damp"], atol=o["atol"], btol=o["btol"],
                                                 conlim=o["conlim"], itnlim=o["itnlim"], wantse=o["wantse"])
    x = (r.x.cpu().numpy() if hasattr(r.x, "cpu") else np.asarray(r.x)).astype(np.float64)
    se = None if r.se is None else (r.se.cpu().numpy() if hasattr(r.se, "cpu") else np.asarray(r.se)).astype(np.float64)
    np.savez(out, x=x, se=se if se is not None else np.zeros(0), istop=r.istop, itn=r.itn, anorm=r.anorm,
             acond=r.acond, rnorm=r.rnorm, arnorm=r.arnorm, xnorm=r.xnorm, row0=row0, nrows=nrows)
    dist.barrier()
    dist.destroy_process_group()


def engine_rank(rank, world, port, case, backend, out):
    """The C++ engine (csrc/shard_engine.h) on this rank's row block, RCCL between the ranks."""
    os.environ["LSQR_RANKS_SHARE_GPU"] = "1"
    real32 = "engine32" in backend
    if backend.endswith("_ov") or backend.endswith("_ov_copy"):
        os.environ["LSQRHIP_SHARD_OVERLAP"] = "1"
        os.environ["LSQRHIP_SHARD_WORLD"] = str(world)
    if "csb" in backend:                 # the blocks in column-swept layouts (with the overlap: built for the parts, the
        os.environ["LSQRHIP_CSB"] = "1"  # products run phase by phase)
    if backend.endswith("_copy"):      # the n-vector exchanges as copy-engine pulls from IPC-mapped peer buffers
        os.environ["LSQRHIP_SHARD_COPY"] = "1"
    from lsqr_amd.dist_bench import share_one_gpu
    assert share_one_gpu(rank)
    import torch
    import torch.distributed as dist
    from cases import build_cases
    from lsqr_amd import capi
    from lsqr_amd.dist import EngineSolver, local_block, partition_rows
    from lsqr_amd.solver import lsqr_solver_ez
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            device_id=torch.device("cuda:0"))
    p, o = build_cases()[case]
    if len(sys.argv) > 7:     # ITNCAP: the case with its iteration limit lowered (what the socket transport between
        o = dict(o, itnlim=min(o["itnlim"], int(sys.argv[7])))   # processes sharing the GPU costs per iteration adds up)
    w = np.bincount(p.irow - 1, minlength=p.m).astype(np.float64)
    row0, nrows = partition_rows(p.m, world, w)[rank]
    irow, icol, a, b = local_block(p.irow, p.icol, p.a, p.b, row0, nrows)
    s = lsqr_solver_ez().initialize(nrows, p.n, a, irow, icol, real32=real32)
    wp = np.float32 if real32 else np.float64
    d_b = capi.DeviceBuffer.from_array((b if nrows else np.zeros(1)).astype(wp))
    eng = EngineSolver(s, row0, p.m, world, rank)
    if backend.endswith("_copy"):
        assert s.get_option("shard_copy") == 1, "LSQRHIP_SHARD_COPY=1 did not take (IPC handles refused?)"
        assert s.get_option("shard_overlap") == (1 if "_ov" in backend else 0)
    r = eng.solve(d_b.ptr.value, damp=o["damp"], atol=o["atol"], btol=o["btol"], conlim=o["conlim"], itnlim=o["itnlim"],
                  wantse=o["wantse"])
    r2 = eng.solve(d_b.ptr.value, damp=o["damp"], atol=o["atol"], btol=o["btol"], conlim=o["conlim"], itnlim=o["itnlim"],
                   wantse=o["wantse"])       # the communicators, streams and buffers serve a second solve: same bits
    x = r.x.to_array(wp, p.n).astype(np.float64)
    se = r.se.to_array(wp, p.n).astype(np.float64) if r.se is not None else np.zeros(0)
    np.savez(out, x=x, se=se, istop=r.istop, itn=r.itn, anorm=r.anorm, acond=r.acond, rnorm=r.rnorm, arnorm=r.arnorm,
             xnorm=r.xnorm, row0=row0, nrows=nrows, again_same=int((r2.istop, r2.itn, r2.anorm, r2.rnorm) ==
                                                                     (r.istop, r.itn, r.anorm, r.rnorm)))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
