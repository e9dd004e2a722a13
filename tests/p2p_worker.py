"""One rank of the device-side (p2p) transport as a PROCESS of its own (tests/test_gpu_p2p.py starts `world` of these on the
box's one GPU: hipIpc works between processes on one device).  python tests/p2p_worker.py <transport: p2p|rccl> <case> <out prefix>
with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (gloo: only the launcher's part -- handing the
mailboxes' handles round -- and the final barrier go through it)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402

from cora_amd import capi, host  # noqa: E402
from cora_amd.dist import NativeP2PComm, NativeRcclComm  # noqa: E402


def problem(n, p, precond, loops=4):
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=5, n_ranges=n // 2, n_loops=loops, seed=11, precond=precond)
    P.update()
    P.set_rank(p)
    return P


CASES = {"small": (900, 4), "mid": (6000, 5), "big": (60000, 4)}   # poses, relaxation rank


def main():
    transport, case, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, p = CASES[case.split("_")[0]]
    chol = case.endswith("chol")
    P = problem(n, p, capi.PRECOND_REGULARIZED_CHOLESKY if chol else capi.PRECOND_JACOBI)
    dm = P.dims()
    make = (lambda ctx: NativeP2PComm(ctx)) if transport == "p2p" else (lambda ctx: NativeRcclComm(ctx))
    comm = P.set_partition(rank, world, make)
    rng = np.random.default_rng(5)
    Y = P.op("projectToManifold", rng.uniform(-1, 1, (dm["N"], p)))
    V = P.op("tangent_space_projection", Y, rng.uniform(-1, 1, (dm["N"], p)))
    c0 = comm.counters()
    f = P.op("evaluateObjective", Y)
    H = P.op("Riemannian_Hessian_vector_product", Y, P.op("Euclidean_gradient", Y), V)
    res = P.tnt(Y, max_iterations=6)
    ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    path = ctx.stpcg_path()
    c1 = comm.counters()
    st = comm.status() if transport == "p2p" else {}
    np.savez(prefix + ".rank%d.npz" % rank, f=f, H=H, x=res["x"], tf=res["f"], iterations=res["iterations"], hvps=res["hvps"],
             path=path, counters=np.array([c1[0] - c0[0], c1[1] - c0[1]]), exchanged=comm.exchanged_rows,
             status=np.array([st.get(k, -1) for k in ("collectives", "kernels", "timeouts", "memory_kind", "allgathers", "allreduces")]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
