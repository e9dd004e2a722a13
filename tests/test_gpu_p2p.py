"""The device-side transport of a partitioned handle (cora_comm_create_p2p, cora_amd/csrc/p2p.h; SURVEY 8e mitigation 3): every
rank a PROCESS of its own, 2 and 4 of them sharing the box's one GPU -- mailboxes exported with hipIpcGetMemHandle and mapped
by the peers, collectives as single kernels that push, set per-peer sequence flags and spin on their own.  Required: the same
BITS as the in-process `local` transport (and so as RCCL: both add in rank order), no library collective on the data path
(cora_comm_counters 0 + 0), no timeout raised.  Reference call sites that run sharded: src/CORA.cpp:139-140,
src/CORA_utils.cpp:83."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from cora_amd import capi, host
from cora_amd.dist import NativeLocalComm
from test_gpu_sharded import _run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, transport, case, prefix, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               HSA_ENABLE_IPC_MODE_LEGACY="0", CORA_P2P_TIMEOUT_S="30")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), transport, case, prefix],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=timeout)[0])
    finally:
        for pr in procs:     # exactly the processes started here
            if pr.poll() is None:
                pr.kill()
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
    return [np.load(prefix + ".rank%d.npz" % r) for r in range(world)]


def _local_reference(world, case):
    import p2p_worker as w
    n, p = w.CASES[case.split("_")[0]]
    chol = case.endswith("chol")
    pre = capi.PRECOND_REGULARIZED_CHOLESKY if chol else capi.PRECOND_JACOBI
    P1 = w.problem(n, p, pre)
    dm = P1.dims()
    rng = np.random.default_rng(5)
    Y = P1.op("projectToManifold", rng.uniform(-1, 1, (dm["N"], p)))
    V = P1.op("tangent_space_projection", Y, rng.uniform(-1, 1, (dm["N"], p)))

    def body(r, group):
        P = w.problem(n, p, pre)
        P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
        f = P.op("evaluateObjective", Y)
        H = P.op("Riemannian_Hessian_vector_product", Y, P.op("Euclidean_gradient", Y), V)
        res = P.tnt(Y, max_iterations=6)
        ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        return f, H, res, ctx.stpcg_path()

    return _run_ranks(world, body, "native")


@pytest.mark.parametrize("world,case", [(2, "small"), (4, "mid"), (4, "mid_chol"), (2, "big_chol")])
def test_p2p_processes_return_the_local_transport_s_bits(world, case, tmp_path):
    """Operators, TNT with the device-resident STPCG (fused vector passes with Jacobi; sweep-fused on every rank's own
    block-Jacobi Cholesky factor with `chol`): exchange of the operand before every product and the iteration's 1 + 2
    reductions all through the mailboxes.  big_chol: 30 000 poses per rank, where a rank's factor is a two-stage solve plan and
    the iteration is the sweep-fused one (stpcg path 2)."""
    got = _launch(world, "p2p", case, str(tmp_path / "p2p"))
    ref = _local_reference(world, case)
    for r in range(world):
        g, (f, H, res, path) = got[r], ref[r]
        assert int(g["path"]) == path and path == (2 if case == "big_chol" else 1)
        assert float(g["f"]) == f
        assert np.array_equal(g["H"], H)
        assert int(g["iterations"]) == res["iterations"] and int(g["hvps"]) == res["hvps"]
        assert float(g["tf"]) == res["f"] and np.array_equal(g["x"], res["x"])
        assert list(g["counters"]) == [0, 0]            # no library collective on the data path
        st = g["status"]
        assert st[2] == 0, "a waiting kernel timed out"
        assert st[4] > 0 and st[5] > 0                  # ... because the mailboxes carried them: all-gathers, all-reduces
        assert 0 < int(g["exchanged"]) < H.shape[0]
    print("\n  %d processes, %s: %d all-gathers + %d all-reduces in %d kernels through the mailboxes (memory kind %d), stpcg path %d"
          % (world, case, got[0]["status"][4], got[0]["status"][5], got[0]["status"][1], got[0]["status"][3], int(got[0]["path"])))


@pytest.mark.parametrize("world,case", [(2, "implicit"), (4, "implicit"), (2, "staircase")])
def test_p2p_processes_on_the_implicit_formulation_and_the_staircase(world, case, tmp_path):
    """The transport's generic collectives under load: the implicit formulation's packed gather of the translation rows (tens of
    kilobytes per rank: several blocks of one all-gather kernel, each handing its own slice over) and the in-place gather of whole
    shards behind every download; solveCORA with certification on a partition (LOBPCG's Gram matrices all-reduced, the sharded
    certificate operator).  The local transport's bits again, and no library collective."""
    import p2p_worker as w
    got = _launch(world, "p2p", case, str(tmp_path / "p2p"))

    def body(r, group):
        _, out = (w.implicit_case if case == "implicit" else w.staircase_case)(lambda ctx: NativeLocalComm(ctx, group), r, world)
        return out

    ref = _run_ranks(world, body, "native")
    for r in range(world):
        g = got[r]
        for k, v in ref[r].items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(g[k], v), (case, r, k)
            else:
                assert float(g[k]) == float(v), (case, r, k, float(g[k]), v)
        assert list(g["counters"]) == [0, 0] and g["status"][2] == 0
        assert g["status"][4] > 0 and g["status"][5] > 0
    print("\n  %d processes, %s: %d all-gathers + %d all-reduces in %d kernels through the mailboxes; rows gathered %d"
          % (world, case, got[0]["status"][4], got[0]["status"][5], got[0]["status"][1], int(got[0]["gathered"])))


def test_a_dead_peer_raises_a_timeout_instead_of_hanging(tmp_path):
    """One rank of two never shows up for the first collective after the set-up: the other's waiting kernel gives up after
    CORA_P2P_TIMEOUT_S, the status reports it (the GPU is not hung), and the NEXT collective call of that rank fails with an error
    instead of computing on what the timed-out collective delivered (the kernels count their timeouts in pinned host memory too,
    which the library reads without a synchronisation)."""
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch.distributed as dist
from cora_amd import capi, host
from cora_amd.dist import NativeP2PComm
sys.path.insert(0, os.path.join(%r, "tests"))
import p2p_worker as w
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
P = w.problem(900, 4, capi.PRECOND_JACOBI)
comm = P.set_partition(rank, 2, lambda ctx: NativeP2PComm(ctx))
dist.barrier()
if rank == 1:
    time.sleep(15)      # "dead": no collective call
    print("STATUS idle")
else:
    dm = P.dims()
    Y = P.op("projectToManifold", np.random.default_rng(1).uniform(-1, 1, (dm["N"], 4)))
    t0 = time.time()
    raised = ""
    try:
        f = P.op("evaluateObjective", Y)     # exchange + all-reduce: the peer never pushes
    except Exception as e:                   # the collective AFTER the one that timed out refuses to compute on
        raised = str(e)
    print("STATUS timeouts=%%d seconds=%%.1f raised=%%r" %% (comm.status()["timeouts"], time.time() - t0, raised))
dist.barrier()
''' % (ROOT, ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
               CORA_P2P_TIMEOUT_S="3")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=300)[0])
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    line = [l for l in outs[0].splitlines() if l.startswith("STATUS")]
    assert line, outs[0][-3000:]
    assert "timeouts=0" not in line[0], line[0]
    assert "did not arrive" in line[0], line[0]      # ... and the next collective call failed loudly
    secs = float(line[0].split("seconds=")[1].split()[0])
    assert secs < 40, line[0]
