"""One rank of a sharded solve (launched by tests/test_dist.py as a subprocess).

    python tests/dist_worker.py RANK WORLD PORT CASE BACKEND OUT.npz

BACKEND = numpy (CPU, gloo, oracle-backed stage stand-in) | hip (C-ABI stages on cuda:0, gloo)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)      # a hang prints where and ends the rank
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    case, backend, out = sys.argv[4], sys.argv[5], sys.argv[6]
    import torch.distributed as dist
    from cases import build_cases
    from lsqr_amd.dist import ShardedLSQR, TorchComm, local_block, partition_rows

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
        s = lsqr_solver_ez().initialize(nrows, p.n, a, irow, icol)
        be = HipShardBackend(s, p.m, world, rank)
        d_b = capi.DeviceBuffer.from_array(b if nrows else np.zeros(1))
        b_arg = d_b.ptr.value
    r = ShardedLSQR(be, comm, poll_every=3).solve(b_arg, damp=o["damp"], atol=o["atol"], btol=o["btol"],
                                                 conlim=o["conlim"], itnlim=o["itnlim"], wantse=o["wantse"])
    x = r.x.cpu().numpy() if hasattr(r.x, "cpu") else np.asarray(r.x)
    se = None if r.se is None else (r.se.cpu().numpy() if hasattr(r.se, "cpu") else np.asarray(r.se))
    np.savez(out, x=x, se=se if se is not None else np.zeros(0), istop=r.istop, itn=r.itn, anorm=r.anorm,
             acond=r.acond, rnorm=r.rnorm, arnorm=r.arnorm, xnorm=r.xnorm, row0=row0, nrows=nrows)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
