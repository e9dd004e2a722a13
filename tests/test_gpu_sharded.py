"""The sharded solver on ONE GPU: every partition of the graph is a rank of a cora_amd.dist.ThreadGroup (one
thread and one partitioned handle each; the exchange is device-to-device row copies).  With the communication
injected (include/cora_hip.h, cora_set_comm) the C ABI's resident entry points are collective and the C++ host
-- operators, TNT, the certificate operator -- runs unchanged on every rank.  Checked against the single-handle
run and the CPU oracle.  Reference call sites that run sharded this way: src/CORA.cpp:139-140 (TNT with the
closures of :52-92,119-122), src/CORA_utils.cpp:83 (the certificate operator)."""
import threading

import numpy as np
import pytest

from cora_amd import capi, host
from cora_amd.dist import NativeLocalComm, NativeLocalGroup, NativeRcclComm, ThreadComm, ThreadGroup
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


# the two in-process transports: Python callbacks (cora_set_comm) and the library's own (cora_comm_create_local)
TRANSPORTS = {"callbacks": (ThreadGroup, ThreadComm), "native": (NativeLocalGroup, NativeLocalComm)}


def _run_ranks(world, body, transport="callbacks"):
    group = TRANSPORTS[transport][0](world)
    out, err = [None] * world, [None] * world

    def run(r):
        try:
            out[r] = body(r, group)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            group.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(900)
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out


def _problem(n, p, seed=11, loops=4):
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=5, n_ranges=n // 2, n_loops=loops, seed=seed,
                               precond=capi.PRECOND_JACOBI)
    P.update()
    P.set_rank(p)
    return P


@pytest.mark.parametrize("transport", ["callbacks", "native"])
@pytest.mark.parametrize("world,n", [(2, 900), (4, 3000)])
def test_sharded_operators_and_tnt_match_single_handle(world, n, transport):
    p = 4
    Comm = TRANSPORTS[transport][1]
    P1 = _problem(n, p)
    dm = P1.dims()
    _, _, rowptr, colidx, vals = P1.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    rng = np.random.default_rng(5)
    Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dm["N"], p)))
    V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, (dm["N"], p)))
    G = orc.egrad(Q, Y)
    ref_hvp = orc.hvp(Q, dims, Y, G, V)
    single = P1.tnt(Y, max_iterations=6, host_stpcg=True)

    def body(r, group):
        P = _problem(n, p)
        comm = P.set_partition(r, world, lambda ctx: Comm(ctx, group))
        assert 0 < comm.exchanged_rows < dm["N"]
        f = P.op("evaluateObjective", Y)
        H = P.op("Riemannian_Hessian_vector_product", Y, P.op("Euclidean_gradient", Y), V)
        # callbacks: the host-driven STPCG (collective calls);  native: the device-resident fused iteration, its inner
        # products all-reduced on the device (cora_stpcg_device_ok)
        res = P.tnt(Y, max_iterations=6)
        ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        assert ctx.stpcg_path() == (1 if transport == "native" else 0)
        return f, H, res

    outs = _run_ranks(world, body, transport)
    for f, H, res in outs:
        assert abs(f - orc.cost(Q, Y)) < 1e-11 * abs(f)
        assert np.abs(H - ref_hvp).max() < 1e-10 * np.abs(ref_hvp).max()
        assert res["iterations"] == single["iterations"] and res["hvps"] == single["hvps"]
        assert abs(res["f"] - single["f"]) < 1e-10 * abs(single["f"])
        assert np.abs(res["x"] - single["x"]).max() < 1e-8
    # every rank returns the same bits (the reductions add in rank order on every rank)
    for f, H, res in outs[1:]:
        assert f == outs[0][0] and np.array_equal(res["x"], outs[0][2]["x"])


@pytest.mark.parametrize("world,n", [(2, 900), (4, 6000)])
def test_exchange_overlaps_the_interior_slices_with_the_same_numbers(world, n):
    """With the library's own communication a product runs as two launches around the exchange of its operand: interior
    slices (rows of the rank's own shard only) while pack / all-gather / scatter are under way on a second stream,
    boundary slices when the remote rows have landed (capi.hip, exchange_and_product).  Every slice computes
    what it computed before, so operators are the serial ones bit for bit; kappa adds its per-block partials in another
    order, so the device-resident STPCG agrees to rounding."""
    p = 5
    P1 = _problem(n, p)
    dm = P1.dims()
    rng = np.random.default_rng(11)
    Y = P1.op("projectToManifold", rng.uniform(-1, 1, (dm["N"], p)))
    V = P1.op("tangent_space_projection", Y, rng.uniform(-1, 1, (dm["N"], p)))
    X = rng.uniform(-1, 1, (dm["N"], p + 2))

    def body(r, group):
        P = _problem(n, p)
        comm = P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
        out = {}
        assert not comm.overlap_active()  # default: these shards are too small for a launch of their own to pay
        for mode in (True, False):
            comm.overlap(2 if mode else 0)
            assert comm.overlap_active() == mode
            H = P.op("Riemannian_Hessian_vector_product", Y, P.op("Euclidean_gradient", Y), V)
            f = P.op("evaluateObjective", Y)
            res = P.tnt(Y, max_iterations=5)
            out[mode] = (f, H, res)
        return out

    outs = _run_ranks(world, body, "native")
    for o in outs:
        assert o[True][0] == o[False][0]
        assert np.array_equal(o[True][1], o[False][1])
        a, b = o[True][2], o[False][2]
        assert a["iterations"] == b["iterations"] and a["hvps"] == b["hvps"]
        assert abs(a["f"] - b["f"]) < 1e-12 * abs(b["f"]) and np.abs(a["x"] - b["x"]).max() < 1e-9


@pytest.mark.parametrize("transport", ["callbacks", "native"])
def test_eight_partitions_of_the_headline_graph(transport):
    """BASELINE config 4 / 5 on one GPU: the 10^5-pose graph cut into 8 row partitions, the Hessian-vector
    product at rank 5 and the certificate operator (Q - Lambda) X with 10 columns against the CPU oracle."""
    world, n, p = 8, 100000, 5
    Comm = TRANSPORTS[transport][1]
    P1 = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_JACOBI)
    P1.update()
    dm = P1.dims()
    _, _, rowptr, colidx, vals = P1.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    rng = np.random.default_rng(7)
    Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dm["N"], p)))
    V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, (dm["N"], p)))
    X10 = rng.uniform(-1, 1, (dm["N"], 10))
    G = orc.egrad(Q, Y)
    ref_hvp = orc.hvp(Q, dims, Y, G, V)
    Lst, lob = orc.lambda_blocks(Q, dims, Y)
    ref_S = orc.S_apply(Q, dims, Lst, lob, X10)
    del P1

    def body(r, group):
        P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_JACOBI)
        P.update()
        P.set_rank(p)
        comm = P.set_partition(r, world, lambda ctx: Comm(ctx, group))
        ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        y, x, o = ctx.dev_alloc(p), ctx.dev_alloc(p), ctx.dev_alloc(p)
        x10, o10 = ctx.dev_alloc(10), ctx.dev_alloc(10)
        ctx.upload(Y, y)
        ctx.set_point_dev(y)
        ctx.upload(V, x)
        ctx.hvp_dev(x, o)
        H = ctx.download(o, p)
        ctx.upload(X10, x10)
        ctx.certificate_product_dev(x10, 10, o10)
        S = ctx.download(o10, 10)
        return comm.exchanged_rows, ctx.rows, H, S

    outs = _run_ranks(world, body, transport)
    for exch, rows, H, S in outs:
        assert exch < rows // 100   # the chain halo + the landmarks' own rows (distributed long rows): < 1 % of the vector
        assert np.abs(H - ref_hvp).max() < 1e-10 * np.abs(ref_hvp).max()
        assert np.abs(S - ref_S).max() < 1e-10 * np.abs(ref_S).max()


def test_rccl_transport_world_one():
    """The RCCL transport on the one GPU of the test box: librccl.so is opened at run time, a communicator of one rank is
    created by the library (cora_rccl_unique_id + cora_comm_create_rccl) and the collective entry points run through
    it (world 1: nothing to exchange, the reductions are identities) -- the plumbing the 8-GPU bench relies on."""
    n, p = 900, 4
    P = _problem(n, p)
    dm = P.dims()
    ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    comm = NativeRcclComm(ctx)
    assert comm.world == 1 and comm.exchanged_rows >= 0
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    rng = np.random.default_rng(5)
    Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dm["N"], p)))
    assert abs(P.op("evaluateObjective", Y) - orc.cost(Q, Y)) < 1e-11 * abs(orc.cost(Q, Y))


@pytest.mark.parametrize("transport", ["callbacks", "native"])
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_staircase_with_certification_matches_single_handle(world, transport):
    """solveCORA on partitioned handles (round-2 advice: it used to throw at its first certification -- 'host Lambda
    blocks need a 1-GPU handle').  Every rank runs the whole staircase of src/CORA.cpp:26-243 in step: TNT, the
    certificate matrix from the gathered Lambda blocks (src/CORA_problem.cpp:1105-1166), the PSD test by factorisation,
    LOBPCG on the sharded certificate operator when it fails (src/CORA_utils.cpp:83-119), the saddle escape, rounding.
    Same decision, same rank, cost equal to the single-handle run's to 1e-8."""
    n = 600
    Comm = TRANSPORTS[transport][1]

    def make():
        P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n, seed=23, precond=capi.PRECOND_JACOBI)
        P.update()
        return P
    P1 = make()
    dm = P1.dims()
    _, _, rowptr, colidx, vals = P1.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    x0 = P1.op("getRandomInitialGuess")
    single = P1.solve(x0, max_rank=8, max_seconds=120)
    cert_single = P1.certify(single["x"], 1e-5)

    def body(r, group):
        P = make()
        P.set_partition(r, world, lambda ctx: Comm(ctx, group))
        res = P.solve(x0, max_rank=8, max_seconds=120)
        cert = P.certify(res["x"], 1e-5)
        return res, cert

    outs = _run_ranks(world, body, transport)
    for res, cert in outs:
        assert res["final_rank"] == single["final_rank"] and res["certified"] == single["certified"]
        assert abs(res["f"] - single["f"]) < 1e-8 * max(1.0, abs(single["f"]))
        assert abs(orc.cost(Q, res["x"]) - res["f"]) < 1e-9 * max(1.0, abs(res["f"]))
        assert np.abs(res["x"] - orc.project_manifold(dims, res["x"])).max() < 1e-9
        assert cert["is_certified"] == cert_single["is_certified"]
        if not cert["is_certified"]:
            assert cert["theta"] < -0.5e-5
    for res, _ in outs[1:]:   # every rank returns the same bits
        assert np.array_equal(res["x"], outs[0][0]["x"]) and res["hvps"] == outs[0][0]["hvps"]


def test_eight_partitions_tnt_and_certification_at_full_size():
    """The solver itself on 8 partitions of the 10^5-pose graph (BASELINE configs 4 and 5), library-native communication:
    TNT (device-resident STPCG, Jacobi) and certify_solution -- Lambda blocks gathered, PSD test, LOBPCG on the sharded
    certificate operator with 10 columns -- against the single-handle run: cost to 1e-8, same certificate decision and
    sign."""
    world, n, p = 8, 100000, 5

    def make():
        P, gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                                       precond=capi.PRECOND_JACOBI, ground_truth=True)
        P.update()
        P.set_rank(p)
        return P, gt
    P1, gt = make()
    Y0 = P1.op("projectToManifold", np.hstack([gt, np.zeros((gt.shape[0], p - gt.shape[1]))]))
    single = P1.tnt(Y0, max_iterations=4)
    eta = 1e-1
    cert1 = P1.certify(single["x"], eta, nx=10)
    # a point that is NOT certifiable (random on the manifold): the PSD test fails and LOBPCG runs -- on the sharded
    # certificate operator with 10 columns in the partitioned run (BASELINE config 5)
    Yr = P1.op("projectToManifold", np.random.default_rng(3).uniform(-1, 1, Y0.shape))
    cert1r = P1.certify(Yr, 1e-3, nx=10)
    assert not cert1r["is_certified"] and cert1r["theta"] < -0.5e-3
    del P1

    def body(r, group):
        P, _ = make()
        comm = P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
        res = P.tnt(Y0, max_iterations=4)
        dm = P.dims()
        ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        path = ctx.stpcg_path()
        cert = P.certify(res["x"], eta, nx=10)
        certr = P.certify(Yr, 1e-3, nx=10)
        assert not certr["is_certified"] and certr["theta"] < -0.5e-3 and abs(np.linalg.norm(certr["x"]) - 1) < 1e-8
        return res, cert, path, comm.exchanged_rows, ctx.rows

    outs = _run_ranks(world, body, "native")
    for res, cert, path, exch, rows in outs:
        assert path == 1
        assert res["hvps"] == single["hvps"] and res["iterations"] == single["iterations"]
        # four outer iterations from a non-stationary start, not a converged value (SURVEY 8c asks 1e-8 on converged f:
        # the staircase test above): the landmark rows are sums of 8 partial sums here and of 40 chunks on one handle,
        # and the trust-region iterates amplify that rounding -- 2e-8 observed; the operators themselves agree to 1e-10
        # (test_eight_partitions_of_the_headline_graph)
        assert abs(res["f"] - single["f"]) < 1e-7 * abs(single["f"])
        assert cert["is_certified"] == cert1["is_certified"]
        if not cert1["is_certified"]:
            assert cert["theta"] < -eta / 2 and cert1["theta"] < -eta / 2
    print("\n8 partitions at 10^5 poses: f=%.6f (single %.6f), %d Hvps, certified=%s theta=%.3e (single %.3e), %d of %d rows exchanged"
          % (outs[0][0]["f"], single["f"], outs[0][0]["hvps"], outs[0][1]["is_certified"], outs[0][1]["theta"], cert1["theta"],
             outs[0][3], outs[0][4]))


@pytest.mark.parametrize("world", [2, 4])
def test_block_jacobi_cholesky_on_partitions(world):
    """RegularizedCholesky on a partitioned Problem = block Jacobi over the ranks: every rank factorises the diagonal
    block of ITS rows of Q + lambda I (same ordering, factorisation and device solve plan as the single-GPU
    preconditioner, src/CORA_problem.cpp:544-614 per shard) and applies it to its own rows.  Checked against a dense
    solve of each rank's block; TNT with it (device-resident STPCG, the
    solve between the residual update and the projection) reaches the single handle's converged cost."""
    n, p = 500, 4

    gt_box = []

    def make(precond):
        P, gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n, seed=31, precond=precond, ground_truth=True)
        P.update()
        P.set_rank(p)
        gt_box.append(gt)
        return P
    P1 = make(capi.PRECOND_REGULARIZED_CHOLESKY)
    dm = P1.dims()
    _, _, rowptr, colidx, vals = P1.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    lam = P1.precond_info()["lam"]
    rng = np.random.default_rng(9)
    Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dm["N"], p)))
    V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, (dm["N"], p)))
    # TNT from where a front end would leave the problem (the generator's ground truth): both preconditioners converge
    gt = gt_box[0]
    Y0 = P1.op("projectToManifold", np.hstack([gt, np.zeros((gt.shape[0], p - gt.shape[1]))]))
    single = P1.tnt(Y0)
    A = Q.to_scipy().toarray() + lam * np.eye(dm["N"])

    def body(r, group):
        P = make(capi.PRECOND_REGULARIZED_CHOLESKY)
        P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
        ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        m = ctx.row_map()
        mine = (m >= ctx.shard_begin) & (m < ctx.shard_begin + ctx.shard_rows)
        info = P.precond_info()
        PV = P.op("precondition", V)     # Problem::precondition: the solve alone (src/CORA_problem.cpp:869-903)
        res = P.tnt(Y0)
        return mine, info["lam"], PV, res, ctx.stpcg_path()

    outs = _run_ranks(world, body, "native")
    ref = np.zeros_like(V)
    for mine, lam_r, PV, res, path in outs:
        assert abs(lam_r - lam) < 1e-9 * lam and path in (1, 2)   # (2: the shard's plan is two-stage, sweep-fused form)
        idx = np.flatnonzero(mine)
        if mine[dm["N"] - 1]:
            idx = idx[idx != dm["N"] - 1]          # the pinned variable stays zero (src/CORA_preconditioners.cpp:78-79)
        ref[idx] = np.linalg.solve(A[np.ix_(idx, idx)], V[idx])
    for mine, lam_r, PV, res, path in outs:
        assert np.abs(PV - ref).max() < 1e-9 * np.abs(ref).max()   # the collective download returns every rank's rows
        # both stop on TNT's relative-decrease test (1e-6): the converged costs agree to that
        assert abs(res["f"] - single["f"]) < 1e-5 * abs(single["f"])
    print("\nblock-Jacobi Cholesky, %d ranks: f=%.8f in %d products (one factor: f=%.8f in %d)"
          % (world, outs[0][3]["f"], outs[0][3]["hvps"], single["f"], single["hvps"]))


@pytest.mark.parametrize("world,n", [(2, 900), (4, 3000)])
def test_collectives_per_product_and_per_stpcg_iteration(world, n):
    """What crosses the ranks, counted (cora_comm_counters): a product on a partitioned handle is ONE collective -- an
    all-gather that carries the exported rows of the operand AND the distributed long rows' partial sums (the owners add
    them in rank order; no memset, no all-reduce) -- and an iteration of the device-resident STPCG is that all-gather
    plus two all-reduces (kappa | <r, r> and <r, v>), the two synchronisation points of preconditioned CG.
    Call sites that run sharded this way: src/CORA.cpp:139-140, src/CORA_utils.cpp:83,113-119."""
    p = 4
    P1 = _problem(n, p)
    dm = P1.dims()
    rng = np.random.default_rng(3)
    Y = P1.op("projectToManifold", rng.uniform(-1, 1, (dm["N"], p)))
    V = P1.op("tangent_space_projection", Y, rng.uniform(-1, 1, (dm["N"], p)))
    ref = P1.op("Riemannian_Hessian_vector_product", Y, P1.op("Euclidean_gradient", Y), V)

    def body(r, group):
        P = _problem(n, p)
        comm = P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
        ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        assert len(ctx.long_rows()) > 0  # the landmark rows are distributed: their slots ride on the exchange
        y, v, o = ctx.dev_alloc(p), ctx.dev_alloc(p), ctx.dev_alloc(p)
        ctx.upload(Y, y)
        ctx.set_point_dev(y)
        ctx.upload(V, v)
        ctx.sync()
        a0 = comm.counters()
        ctx.hvp_dev(v, o)
        ctx.sync()
        a1 = comm.counters()
        H = ctx.download(o, p)   # (collective: one all-gather of whole shards)
        a2 = comm.counters()
        ctx.spmm_dev(v, p, o)
        ctx.certificate_product_dev(v, p, o)
        ctx.sync()
        a3 = comm.counters()
        # the device-resident STPCG: exactly `it` iterations, nothing else in between
        s, rr, vv, pk, hp, g = (ctx.dev_alloc(p) for _ in range(6))
        ctx.upload(V, g)
        b0 = comm.counters()
        it, _ = ctx.stpcg_dev(g, 1e9, s, rr, vv, pk, hp, max_iters=6)
        ctx.sync()
        b1 = comm.counters()
        assert ctx.stpcg_path() == 1
        ph = comm.product_phases(v, o, epi=2, reps=5)   # the bench's per-phase diagnostic (collective call)
        assert len(ph) == 5 and all(t >= 0.0 for t in ph.values()) and ph["slices_us"] > 0.0
        for q in (y, v, o, s, rr, vv, pk, hp, g):
            ctx.dev_free(q)
        return dict(product=(a1[0] - a0[0], a1[1] - a0[1]), download=(a2[0] - a1[0], a2[1] - a1[1]),
                    two=(a3[0] - a2[0], a3[1] - a2[1]), stpcg=(b1[0] - b0[0], b1[1] - b0[1]), it=it, H=H)

    outs = _run_ranks(world, body, "native")
    for o in outs:
        assert np.abs(o["H"] - ref).max() < 1e-10 * np.abs(ref).max()
        assert o["product"] == (1, 0)
        assert o["two"] == (2, 0)
        it = o["it"]
        assert it >= 1   # (at a random point the Hessian is indefinite: the solve stops on negative curvature early)
        # per iteration 1 all-gather + 2 all-reduces; before the first one the solve applies the preconditioner to the
        # gradient and takes <r, r>, <r, v> (one all-reduce); iterations enqueued past the stopping point (the host runs
        # ahead by a batch of four) still communicate, neutralised by the state
        ag, ar = o["stpcg"]
        assert it <= ag <= it + 4 and ar == 2 * ag + 1, (it, ag, ar)
    print("\nworld %d: product %s, STPCG of %d iterations %s (all-gathers, all-reduces)" % (
        world, outs[0]["product"], outs[0]["it"], outs[0]["stpcg"]))


def test_bench_two_ranks_runs_end_to_end_over_gloo(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), here with both ranks
    on the box's one GPU and CORA_BENCH_BACKEND=gloo: the N > 1 branch of the bench -- partitioned handles, the collective
    set-up, the timed steps with their barrier, the gathered parity check, the multi_gpu block of the JSON line -- runs
    end to end.  (RCCL itself needs one GPU per rank; its one-rank case is test_rccl_transport_at_world_one.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CORA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
           "--poses", "20000", "--pmc-traffic", "off"]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0
    assert d["parity_max_rel_err_vs_cpu"] < 1e-10
    assert "multi_gpu" in d and d["multi_gpu"]["exchanged_rows_per_product"] > 0
    # round 6: the transports are timed in turn -- here the device-side one (two processes sharing the GPU: hipIpc) and the
    # injected torch.distributed callbacks --, every one must give the first one's bits, `value` is the faster one's
    runs = {e["transport"]: e for e in d["multi_gpu"]["transports_timed"]}
    assert set(runs) == {"p2p", "torch"} and all(e["ok"] for e in runs.values()), runs
    assert runs["p2p"]["p2p_status"]["timeouts"] == 0 and runs["p2p"]["library_collectives_so_far"] == [0, 0]
    assert abs(d["ms_per_step"] * 1e3 - min(e["step_us"] for e in runs.values())) < 1e-6 * d["ms_per_step"] * 1e3 + 1e-9
    print("\n  2 ranks on one GPU, 20 000 poses: " + ", ".join("%s %.1f us per step" % (k, e["step_us"]) for k, e in runs.items()))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_products_in_the_window_form(world):
    """The LDS-window form of the chain slices on partitions (windows clipped to the shard's rows, a shard's first pose
    with its predecessor on another rank, remote landmark rows in the tails), serial and overlapped: at the sizes of the
    other tests every shard takes the gather form, so the window form is forced here."""
    L = capi.load()
    old = L.cora_debug_spmm_window_min_slices(0)
    try:
        n, p = 1000, 5
        P1 = _problem(n, p)
        dm = P1.dims()
        _, _, rowptr, colidx, vals = P1.matrix("DataMatrix")
        Q = orc.CSR(rowptr, colidx, vals, dm["N"])
        dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
        rng = np.random.default_rng(9)
        Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dm["N"], p)))
        V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, (dm["N"], p)))
        ref = orc.hvp(Q, dims, Y, orc.egrad(Q, Y), V)
        ref_qx = orc.spmm(Q, V)

        def body(r, group):
            P = _problem(n, p)
            comm = P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
            out = []
            for mode in (0, 2):
                comm.overlap(mode)
                out.append((P.op("Riemannian_Hessian_vector_product", Y, P.op("Euclidean_gradient", Y), V),
                            P.op("Euclidean_gradient", V)))   # (= Q V)
            return out

        for res in _run_ranks(world, body, "native"):
            for H, QX in res:
                assert np.abs(H - ref).max() < 1e-10 * np.abs(ref).max()
                assert np.abs(QX - ref_qx).max() < 1e-10 * np.abs(ref_qx).max()
            assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    finally:
        L.cora_debug_spmm_window_min_slices(old)


@pytest.mark.parametrize("world", [2, 4])
def test_sweep_fused_iteration_on_partitions(world):
    """The sweep-fused STPCG iteration per shard (round-3 review: it only ran on one handle).  Every rank's block-Jacobi
    factor is a two-stage solve plan of its own, so the iteration is: product with the partials of kappa | all-reduce,
    scalar step | forward sweep with r += alpha Hp and the slots of <r, r> and |y|^2 | last stage, whose tail block leaves
    the RANK'S <r, r> and <r, v> = |L_k^-1 r_k|^2 | ONE all-reduce for the two, scalar step | backward sweep with
    v = Proj_Y(x), s += alpha p, p = -v + beta p.  Same iteration as the vector-pass form on the same partition
    (CORA_NO_SWEEP_FUSE=1): same number of products, cost to 1e-9 after ten outer iterations; still two all-reduces
    and one all-gather per iteration."""
    import os
    n, p = 30000 * world, 4

    def make():
        P, gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=6, n_ranges=n // 2, seed=5,
                                       precond=capi.PRECOND_REGULARIZED_CHOLESKY, ground_truth=True)
        P.update()
        P.set_rank(p)
        return P, gt
    P0, gt = make()
    # (a perturbed ground truth: the inner solves then run tens of iterations, not two)
    Y0 = P0.op("projectToManifold", np.hstack([gt, np.zeros((gt.shape[0], p - gt.shape[1]))])
               + 0.05 * np.random.default_rng(2).standard_normal((gt.shape[0], p)))
    del P0

    def body(r, group):
        P, _ = make()
        comm = P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
        dm = P.dims()
        ctx = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        P.precond_info()
        c0 = comm.counters()
        res = P.tnt(Y0, max_iterations=10)
        c1 = comm.counters()
        return res, ctx.stpcg_path(), (c1[0] - c0[0], c1[1] - c0[1])

    out = {}
    for mode, names in {"sweep": (), "passes": ("CORA_NO_SWEEP_FUSE",)}.items():
        for var in names:
            os.environ[var] = "1"
        try:
            out[mode] = _run_ranks(world, body, "native")
        finally:
            for var in names:
                os.environ.pop(var, None)
    for (a, pa, ca), (b, pb, cb) in zip(out["sweep"], out["passes"]):
        assert pa == 2 and pb == 1
        assert a["hvps"] == b["hvps"] and a["iterations"] == b["iterations"] and a["hvps"] >= 15
        assert abs(a["f"] - b["f"]) < 1e-9 * abs(b["f"])
        assert ca == cb   # the same collectives: the form changes what a rank does between them
    print("\nsweep-fused on %d partitions: f=%.8f in %d products (vector passes: %.8f in %d)"
          % (world, out["sweep"][0][0]["f"], out["sweep"][0][0]["hvps"], out["passes"][0][0]["f"], out["passes"][0][0]["hvps"]))


@pytest.mark.parametrize("world", [2, 4])
def test_ildl_preconditioned_certification_on_partitions(world):
    """Step 3 of fast_verification (src/CORA_utils.cpp:129-167) on a partitioned Problem (round-3 review: the remaining
    budget ran unpreconditioned there).  The incomplete L D L^T is a recurrence over the whole chain, so it becomes block
    Jacobi over the ranks like the Cholesky preconditioner: every rank factorises the diagonal block of S + eta I on ITS
    rows and applies it with the device solve plan.  The point is a few TNT iterations from a slightly perturbed ground
    truth: lambda_min(S) is negative and tiny, the unpreconditioned iteration does not find the direction within the
    budget, the preconditioned one does -- on one handle and on the partitions (more iterations: the coupling between
    the shards is not in the preconditioner), every rank with the same numbers."""
    n, p, eta = 3000, 4, 1e-3

    def make():
        P, gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n, seed=23, precond=capi.PRECOND_JACOBI,
                                       ground_truth=True)
        P.update()
        P.set_rank(p)
        return P, gt
    P1, gt = make()
    Y0 = P1.op("projectToManifold", np.hstack([gt, np.zeros((gt.shape[0], p - gt.shape[1]))])
               + 0.02 * np.random.default_rng(1).standard_normal((gt.shape[0], p)))
    Y = P1.tnt(Y0, max_iterations=12)["x"]
    P1.set_verification_lab(seed=False, ildl=True)
    one = P1.certify(Y, eta, nx=6)
    assert not one["is_certified"] and P1.certification_reached_step3() and one["theta"] < -eta / 2
    P1.set_verification_lab(seed=False, ildl=False)
    plain = P1.certify(Y, eta, nx=6)
    assert P1.certification_reached_step3() and plain["iters"] > 4 * one["iters"]

    def body(r, group):
        P, _ = make()
        P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
        P.set_verification_lab(seed=False, ildl=True)
        c = P.certify(Y, eta, nx=6)
        return c, P.certification_reached_step3()

    outs = _run_ranks(world, body, "native")
    dm = P1.dims()
    _, _, rowptr, colidx, vals = P1.matrix("DataMatrix")
    from certhelp import certificate_matrix
    Sd = certificate_matrix(orc.CSR(rowptr, colidx, vals, dm["N"]), orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"]), Y)
    for c, step3 in outs:
        assert step3 and not c["is_certified"]
        assert c["theta"] < -eta / 2                       # the reference's stopping rule, reached ...
        assert c["iters"] < plain["iters"] // 2            # ... well inside the budget the plain iteration exhausts
        assert abs(np.linalg.norm(c["x"]) - 1) < 1e-8
        assert abs(c["theta"] - c["x"] @ (Sd @ c["x"])) < 1e-8 * max(1.0, abs(c["theta"]))   # curvature by the oracle's S
    for c, _ in outs[1:]:
        assert c["iters"] == outs[0][0]["iters"] and np.array_equal(c["x"], outs[0][0]["x"])
    print("\nILDL-preconditioned LOBPCG: one handle %d iterations, %d partitions %d, unpreconditioned %d (budget)"
          % (one["iters"], world, outs[0][0]["iters"], plain["iters"]))


@pytest.mark.parametrize("transport", ["callbacks", "native"])
@pytest.mark.parametrize("world", [2, 4])
def test_implicit_formulation_on_partitions(world, transport):
    """Formulation::Implicit (src/CORA_problem.cpp:714-753) on a partitioned Problem (round-3 review: it was single
    handle only).  Q_impl Y = Q [Y; t; 0] with t = -M^-1 B^T Y: the two products are the partitioned products, the
    translation solve -- a recurrence over the whole chain -- is replicated (the right-hand side is gathered, every rank
    runs the same solve plan).  The operators against the oracle's Schur-complement restatement, TNT and the
    certification against the single handle (cost to 1e-8, same decision)."""
    n, p = 700, 4
    Comm = TRANSPORTS[transport][1]

    gt_box = []

    def make():
        P, gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n // 2, n_loops=6, seed=19,
                                       precond=capi.PRECOND_JACOBI, ground_truth=True)
        P.update()
        P.set_formulation(True)
        P.set_rank(p)
        gt_box.append(gt)
        return P
    P1 = make()
    dm = P1.dims()
    _, _, rowptr, colidx, vals = P1.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    I = orc.Implicit(Q, dims)
    Y = P1.op("getRandomInitialGuess")
    V = P1.op("tangent_space_projection", Y, np.random.default_rng(3).uniform(-1, 1, Y.shape))
    G1 = P1.op("Euclidean_gradient", Y)
    # TNT from where a front end would leave the problem (the generator's ground truth, rotations and ranges): converges
    m = P1.variable_size()
    Y0 = P1.op("projectToManifold", np.hstack([gt_box[0][:m], np.zeros((m, p - gt_box[0].shape[1]))]))
    single = P1.tnt(Y0)
    cert1 = P1.certify(single["x"], 1e-4)

    def body(r, group):
        P = make()
        P.set_partition(r, world, lambda ctx: Comm(ctx, group))
        f = P.op("evaluateObjective", Y)
        G = P.op("Euclidean_gradient", Y)
        H = P.op("Riemannian_Hessian_vector_product", Y, G, V)
        X = P.op("getTranslationExplicitSolution", Y)
        res = P.tnt(Y0)
        cert = P.certify(res["x"], 1e-4)
        # the library's own communication gathers the TRANSLATION rows of the replicated solve's right-hand side, packed
        # (world x the longest shard's translation rows), not whole shards
        comm = getattr(P, "_comm", None)
        if transport == "native" and comm is not None and hasattr(comm, "gathered_rows"):
            h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
            r0 = comm.gathered_rows()
            P.op("Euclidean_gradient", Y)
            per_product = comm.gathered_rows() - r0   # the operator's one lift (+ whatever a download gathers: whole shards)
            packed = per_product - world * h.shard_rows
            assert (dm["n"] + dm["l"]) <= packed <= 2 * (dm["n"] + dm["l"]) and packed < world * h.shard_rows // 2, (
                per_product, world, h.shard_rows)
        return f, G, H, X, res, cert

    outs = _run_ranks(world, body, transport)
    Gref = I.product(Y)
    scale = np.abs(Gref).max()
    Xref = I.translation_explicit(Y)
    for f, G, H, X, res, cert in outs:
        assert abs(f - I.cost(Y)) < 1e-10 * max(1.0, abs(f))
        assert np.abs(G - Gref).max() < 1e-10 * scale and np.abs(G - G1).max() < 1e-10 * scale
        assert np.abs(H - I.hvp(Y, V)).max() < 1e-10 * scale
        assert np.abs(X - Xref).max() < 1e-9 * max(1.0, np.abs(Xref).max())
        assert abs(res["f"] - single["f"]) < 1e-8 * max(1.0, abs(single["f"]))
        assert cert["is_certified"] == cert1["is_certified"]
        if not cert["is_certified"]:
            assert cert["theta"] < -0.5e-4
    for o in outs[1:]:
        assert np.array_equal(o[4]["x"], outs[0][4]["x"])
