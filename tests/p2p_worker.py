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


def implicit_case(make_comm, rank, world):
    """Formulation::Implicit on a partition: the translation solve is replicated behind a PACKED all-gather of the translation
    rows (cora_native_comm::allgather_rows) -- on this transport several mailbox pieces handed over by a multi-block kernel --,
    and downloads gather whole shards in place."""
    n, p = 2500, 4
    P, gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n // 2, n_loops=6, seed=19,
                                   precond=capi.PRECOND_JACOBI, ground_truth=True)
    P.update()
    P.set_formulation(True)
    P.set_rank(p)
    comm = P.set_partition(rank, world, make_comm)
    Y = P.op("getRandomInitialGuess")
    V = P.op("tangent_space_projection", Y, np.random.default_rng(3).uniform(-1, 1, Y.shape))
    f = P.op("evaluateObjective", Y)
    G = P.op("Euclidean_gradient", Y)
    H = P.op("Riemannian_Hessian_vector_product", Y, G, V)
    X = P.op("getTranslationExplicitSolution", Y)
    m = P.variable_size()
    Y0 = P.op("projectToManifold", np.hstack([gt[:m], np.zeros((m, p - gt.shape[1]))]))
    res = P.tnt(Y0, max_iterations=8)
    return (P, comm), dict(f=f, G=G, H=H, X=X, x=res["x"], tf=res["f"], iterations=res["iterations"], hvps=res["hvps"])


def staircase_case(make_comm, rank, world):
    """solveCORA with certification on a partition (Lambda blocks through the collective download, the sharded certificate
    operator inside LOBPCG, Gram matrices all-reduced)."""
    n = 800
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n // 2, n_loops=3, seed=23, precond=capi.PRECOND_JACOBI)
    P.update()
    P.set_rank(3)
    comm = P.set_partition(rank, world, make_comm)
    x0 = P.op("getRandomInitialGuess")
    res = P.solve(x0, max_rank=6, max_seconds=120)
    return (P, comm), dict(x=res["x"], tf=res["f"], levels=res["levels"], hvps=res["hvps"], certified=int(res["certified"]),
                           final_rank=res["final_rank"])


def main():
    transport, case, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if case in ("implicit", "staircase"):
        make = (lambda ctx: NativeP2PComm(ctx)) if transport == "p2p" else (lambda ctx: NativeRcclComm(ctx))
        # (the Problem owns the handle the communicator lives on: both are kept until the status has been read)
        (P, comm), out = (implicit_case if case == "implicit" else staircase_case)(make, rank, world)
        st = comm.status() if transport == "p2p" else {}
        c = comm.counters()
        np.savez(prefix + ".rank%d.npz" % rank, counters=np.array(c), gathered=comm.gathered_rows(),
                 status=np.array([st.get(k, -1) for k in ("collectives", "kernels", "timeouts", "memory_kind", "allgathers", "allreduces")]), **out)
        dist.barrier()
        dist.destroy_process_group()
        return
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
