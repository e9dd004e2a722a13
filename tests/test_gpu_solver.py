"""GPU parity of the device-resident truncated-Newton solver (cora_amd/csrc/host/TNT.cpp)
against the numpy restatement of the same published algorithm on the oracle operators
(oracle/tnt.py).  Tolerances: converged cost 1e-8 relative (or 1e-9 absolute when the optimum
is 0), both gradient norms below the stopping tolerance -- trajectories of the reference's own
TNT are unpinned (sources absent), see DESIGN.md section 4."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from cora_amd import capi, host
from oracle import oracle as orc
from oracle import tnt as otnt

pytestmark = pytest.mark.gpu


def _oracle_problem(P):
    dm = P.dims()
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    return orc.CSR(rowptr, colidx, vals, dm["N"]), orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])


@pytest.mark.parametrize("p", [2, 3])
def test_tnt_golden_noiseless(p):
    P = host.Problem.from_pyfg(os.path.join(GOLDEN, "small_ra_slam_problem", "factor_graph.pyfg"))
    P.update()
    P.set_preconditioner(capi.PRECOND_JACOBI)
    P.set_rank(p)
    Q, dims = _oracle_problem(P)
    x0 = orc.project_manifold(dims, np.random.default_rng(1).uniform(-1, 1, (dims.N, p)))
    got = P.tnt(x0)
    ref = otnt.tnt(Q, dims, x0)
    assert abs(got["f"] - ref["f"]) < 1e-9  # noiseless data: optimum is 0
    assert got["grad_norm"] < 1e-6 or got["pgrad_norm"] < 1e-6
    # the returned point is on the manifold and its cost is what the solver says
    assert np.abs(got["x"] - orc.project_manifold(dims, got["x"])).max() < 1e-12
    assert abs(orc.cost(Q, got["x"]) - got["f"]) < 1e-9
    assert abs(got["iterations"] - (ref["iterations"] - 1)) <= 3


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("d,n,p,loops", [(3, 150, 3, 0), (3, 150, 5, 6), (2, 200, 3, 5)])
def test_tnt_synthetic_noisy(d, n, p, loops, fused):
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=3, n_ranges=n // 2, n_loops=loops, seed=21)
    P.update()
    P.set_rank(p)
    Q, dims = _oracle_problem(P)
    x0 = orc.project_manifold(dims, np.random.default_rng(2).uniform(-1, 1, (dims.N, p)))
    if not fused:
        os.environ["CORA_NO_FUSE"] = "1"
    try:
        got = P.tnt(x0, max_seconds=120)
    finally:
        os.environ.pop("CORA_NO_FUSE", None)
    ref = otnt.tnt(Q, dims, x0)
    f0 = orc.cost(Q, x0)
    assert got["f"] < 0.05 * f0  # it optimised (the Jacobi preconditioner is weak on chains)
    # Same algorithm, same landscape.  Converged runs agree to 1e-8; runs stopped by the relative-decrease rule or
    # the iteration limit are compared at 1e-4: rounding differences accumulate over ~10^4 Hessian-vector products on
    # this weakly preconditioned chain and move the point where the relative-decrease rule fires (observed: the
    # oracle at the iteration limit, 251 iterations, f = 52.20323; unfused device loop 250 iterations, 52.20347; fused
    # loop -- kappa summed per slice in the product's epilogue -- stops on relative decrease after 217, 52.20079).
    tol = 1e-8 if got["status"] in (0, 1) else 1e-4
    assert abs(got["f"] - ref["f"]) <= tol * abs(ref["f"])
    expected_iters = ref["iterations"] - (ref["status"] in ("gradient", "preconditioned_gradient", "iteration_limit"))
    # Whole runs: same algorithm, same landscape, but 250 outer iterations and ~17 000 products on a weakly
    # preconditioned chain are a chaotic trajectory -- the rounding of the product's sums moves the point where the
    # relative-decrease rule fires (observed 191 .. 251 iterations over the builds of rounds 2-4), so the whole run is held
    # to the same order of work.  What pins the iteration itself is the step-by-step comparison below.
    assert abs(got["iterations"] - expected_iters) <= max(2, 0.3 * expected_iters)
    assert abs(got["hvps"] - ref["hvps"]) <= 0.3 * ref["hvps"] + 2
    rg = orc.rgrad(Q, dims, got["x"])
    assert abs(np.linalg.norm(rg) - got["grad_norm"]) < 1e-6 * max(1.0, got["grad_norm"])
    assert abs(orc.cost(Q, got["x"]) - got["f"]) < 1e-10 * abs(got["f"])


from lockstep import SHORT_SOLVE, lockstep as _lockstep  # noqa: E402,F401


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("d,n,p,loops", [(3, 150, 5, 6), (2, 200, 3, 5)])
def test_tnt_iterations_in_lockstep_with_the_oracle_jacobi(d, n, p, loops, fused):
    """a-T, Jacobi preconditioner (long inner solves): 40 outer iterations, each compared with the oracle's iteration
    from the same point -- device-resident STPCG in both its forms (fused passes / one operation per launch)."""
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=3, n_ranges=n // 2, n_loops=loops, seed=21)
    P.update()
    P.set_rank(p)
    Q, dims = _oracle_problem(P)
    x0 = orc.project_manifold(dims, np.random.default_rng(2).uniform(-1, 1, (dims.N, p)))
    if not fused:
        os.environ["CORA_NO_FUSE"] = "1"
    try:
        worst, steps = _lockstep(P, Q, dims, x0, 40, {})
    finally:
        os.environ.pop("CORA_NO_FUSE", None)
    print("\nlockstep (jacobi, fused=%s): %d iterations, worst deviations %s" % (fused, steps, worst))
    assert steps == 40


@pytest.mark.parametrize("d,n,p,loops", [(3, 400, 4, 0), (2, 600, 3, 10)])
def test_tnt_iterations_in_lockstep_with_the_oracle_cholesky(d, n, p, loops):
    """a-T with the reference's default preconditioner (RegularizedCholesky): 25 outer iterations from a random point in
    lockstep with the oracle.  (Whole runs under this preconditioner: test_tnt_regularized_cholesky -- the converged
    cost agrees, the iteration at which the relative-decrease rule fires does not have to: observed 7 iterations / 129
    products against the oracle's 19 / 1073 from the same start near the minimiser, both ending at the same cost.)"""
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=4, n_ranges=n // 2, n_loops=loops, seed=31,
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    P.set_rank(p)
    Q, dims = _oracle_problem(P)
    lam = P.precond_info()["lam"]
    x0 = orc.project_manifold(dims, np.random.default_rng(5).uniform(-1, 1, (dims.N, p)))
    worst, steps = _lockstep(P, Q, dims, x0, 25, dict(precond="chol", lam=lam))
    print("\nlockstep (cholesky): %d iterations, worst deviations %s" % (steps, worst))
    assert steps == 25


def test_cholesky_preconditioner_apply():
    """blockCholeskySolve semantics (src/CORA_preconditioners.cpp:46-83) on the device's
    staged solve plan (trisolve.h): N-1 leading rows solved, last row zero."""
    import scipy.sparse as sp
    P = host.Problem.synthetic(dim=3, n_poses=900, n_landmarks=5, n_ranges=600, n_loops=8, seed=13,
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    P.set_rank(5)
    Q, dims = _oracle_problem(P)
    info = P.precond_info()
    assert info["lam"] > 0 and info["nnz"] > dims.N
    # lambda_reg = ||Q||_2 / (1e6 - 1), src/CORA_problem.cpp:591 (norm estimated to ~1e-2 like the reference)
    import scipy.sparse.linalg as spl
    lmax = spl.eigsh(Q.to_scipy(), k=1, which="LA", return_eigenvectors=False)[0]
    assert abs(info["lam"] * (1e6 - 1) - lmax) < 0.03 * lmax
    V = np.random.default_rng(3).standard_normal((dims.N, 5))
    out = P.op("precondition", V)
    assert np.all(out[-1] == 0.0)
    M = (Q.to_scipy() + info["lam"] * sp.eye(dims.N)).tocsr()[:dims.N - 1, :dims.N - 1]
    ref = orc.Cholesky(orc.CSR.from_scipy(M)).solve(V[:-1])
    assert np.abs(out[:-1] - ref).max() < 1e-8 * np.abs(ref).max()
    res = M @ out[:-1] - V[:-1]
    assert np.abs(res).max() < 1e-9 * np.abs(V).max() * 10
    # BlockCholesky: documented semantics, three DISTINCT diagonal blocks (note on the
    # reference's coincidentally-passing 2-block test)
    P.set_preconditioner(capi.PRECOND_BLOCK_CHOLESKY)
    out = P.op("precondition", V)
    Qs = Q.to_scipy().tocsr()
    b1, b2 = dims.dn, dims.dn + dims.r
    ref = np.zeros_like(V)
    for lo, hi in ((0, b1), (b1, b2), (b2, dims.N - 1)):
        blk = (Qs[lo:hi, lo:hi] + 1e-3 * sp.eye(hi - lo)).tocsc()
        ref[lo:hi] = spl.spsolve(blk, V[lo:hi])
    assert np.abs(out - ref).max() < 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize("d,n,p,loops", [(3, 400, 4, 0), (2, 600, 3, 10)])
def test_tnt_regularized_cholesky(d, n, p, loops):
    """The reference's default configuration: RegularizedCholesky-preconditioned TNT.  Stage 1 runs
    from a random point (long, chaotic trajectory: only sanity-checked); stage 2 restarts both solvers
    from the same small perturbation of the stage-1 solution, where they must agree to 1e-8."""
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=4, n_ranges=n // 2, n_loops=loops, seed=31,
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    P.set_rank(p)
    Q, dims = _oracle_problem(P)
    x0 = orc.project_manifold(dims, np.random.default_rng(5).uniform(-1, 1, (dims.N, p)))
    f0 = orc.cost(Q, x0)
    s1 = P.tnt(x0, max_seconds=120)
    assert s1["f"] < 1e-4 * f0  # the exact preconditioner gets there (Jacobi stalls at ~3e-2 f0)
    lam = P.precond_info()["lam"]
    xi = orc.tangent_proj(dims, s1["x"], np.random.default_rng(6).standard_normal((dims.N, p)))
    x1 = orc.retract(dims, s1["x"], 1e-3 * xi)
    got = P.tnt(x1, max_seconds=120)
    ref = otnt.tnt(Q, dims, x1, precond="chol", lam=lam)
    # both runs end by the relative-decrease rule (1e-6, src/CORA.cpp:107), which bounds how well two
    # correct implementations can agree: 1e-4 relative here (observed 1e-6 .. 1e-5), 1e-8 when the gradient test fires
    tol = 1e-8 if got["status"] in (0, 1) else 1e-4
    assert abs(got["f"] - ref["f"]) <= tol * abs(ref["f"]) + 1e-9
    assert got["f"] <= orc.cost(Q, x1)
    # near the minimiser the relative-decrease rule makes the exact stopping iteration sensitive to
    # rounding; require the same order of work, not the same count
    assert got["iterations"] <= 2 * ref["iterations"] + 3
    rg = orc.rgrad(Q, dims, got["x"])
    assert abs(np.linalg.norm(rg) - got["grad_norm"]) < 1e-6 * max(1.0, got["grad_norm"])


@pytest.mark.parametrize("d,n,p", [(2, 5000, 2), (3, 12000, 3), (3, 4000, 4), (3, 12000, 5), (2, 6000, 7), (3, 3000, 12),
                                   (3, 12000, 16), (3, 2000, 24)])
def test_cholesky_preconditioner_every_row_stride(d, n, p):
    """The solve kernels are instantiated per row stride (odd strides use scalar loads, even ones 16-byte
    loads): every class of stride, on graphs small enough for the one-stage plan (a single explicit inverse)
    and large enough for dense leaf blocks plus a top stage."""
    import ctypes as C
    import scipy.sparse as sp
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=6, n_ranges=n // 2, n_loops=5, seed=100 + p,
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    P.set_rank(p)
    Q, dims = _oracle_problem(P)
    info = P.precond_info()
    st = (C.c_int64 * 4)()
    capi.load().cora_precond_stats(C.c_void_p(P.context_ptr()), st)
    assert st[0] >= (2 if n >= 12000 else 1) and st[2] == info["nnz"]
    V = np.random.default_rng(p).standard_normal((dims.N, p))
    out = P.op("precondition", V)
    assert np.all(out[-1] == 0.0)
    M = (Q.to_scipy() + info["lam"] * sp.eye(dims.N)).tocsr()[:dims.N - 1, :dims.N - 1]
    res = M @ out[:-1] - V[:-1]
    assert np.abs(res).max() < 1e-8 * np.abs(V).max()


@pytest.mark.parametrize("precond", [capi.PRECOND_JACOBI, capi.PRECOND_REGULARIZED_CHOLESKY])
def test_device_stpcg_matches_host_driven_loop(precond):
    """cora_stpcg_dev keeps the scalar recurrences on the device; the iteration is the same as the loop
    driven from the host one inner product at a time, so TNT must take the same path."""
    P = host.Problem.synthetic(dim=3, n_poses=600, n_landmarks=4, n_ranges=400, n_loops=6, seed=77, precond=precond)
    P.update()
    P.set_rank(4)
    x0 = P.op("getRandomInitialGuess")
    a = P.tnt(x0, max_iterations=12, host_stpcg=False)
    b = P.tnt(x0, max_iterations=12, host_stpcg=True)
    assert a["iterations"] == b["iterations"] and a["status"] == b["status"]
    assert abs(a["hvps"] - b["hvps"]) <= 1
    assert abs(a["f"] - b["f"]) < 1e-9 * abs(b["f"])
    assert np.abs(a["x"] - b["x"]).max() < 1e-7


def test_precondition_in_place_and_out_of_place_agree():
    """cora_precondition_projected_dev(v, v): the staged solve needs distinct right-hand side and output, so
    the in-place call goes through a scratch copy."""
    P = host.Problem.synthetic(dim=3, n_poses=3000, n_landmarks=4, n_ranges=1500, n_loops=4, seed=9,
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    p = 5
    P.set_rank(p)
    P.precond_info()
    dm = P.dims()
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    y, v, o = h.dev_alloc(p), h.dev_alloc(p), h.dev_alloc(p)
    rng = np.random.default_rng(2)
    h.upload(rng.uniform(-1, 1, (dm["N"], p)), y)
    h.project_to_manifold_dev(y, y)
    h.set_point_dev(y)
    h.upload(rng.uniform(-1, 1, (dm["N"], p)), v)
    h.precondition_projected_dev(v, o)
    a = h.download(o, p)
    h.precondition_projected_dev(v, v)
    b = h.download(v, p)
    assert np.array_equal(a, b)
    for q in (y, v, o):
        h.dev_free(q)


@pytest.mark.parametrize("n,precond", [(600, capi.PRECOND_REGULARIZED_CHOLESKY), (600, capi.PRECOND_JACOBI),
                                        (12000, capi.PRECOND_REGULARIZED_CHOLESKY)])
def test_host_loop_forms_take_the_same_path_bit_for_bit(n, precond):
    """How far the host runs ahead of the inner solve's state (one product: the default; nothing; one iteration; batches of
    four) and whether the outer iteration's calls are fused decide WHEN launches are enqueued, never what they compute: the
    solver returns the same bits under every form (explicit-inverse, vector-pass and sweep-fused iterations)."""
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n // 2, n_loops=6, seed=12, precond=precond)
    P.update()
    P.set_rank(4)
    x0 = P.op("getRandomInitialGuess")
    forms = [{}, {"CORA_STPCG_DEPTH": "0"}, {"CORA_STPCG_DEPTH": "1"}, {"CORA_STPCG_BATCH": "4"}, {"CORA_STPCG_BATCH": "1"},
             {"CORA_NO_TNT_FUSE": "1"}, {"CORA_STPCG_AHEAD": "1", "CORA_STPCG_DEPTH": "0"}]
    runs = []
    for env in forms:
        os.environ.update(env)
        try:
            runs.append(P.tnt(x0, max_iterations=15))
        finally:
            for k in env:
                os.environ.pop(k, None)
    a = runs[0]
    assert a["hvps"] > 30
    for env, b in zip(forms[1:], runs[1:]):
        assert (a["iterations"], a["status"], a["hvps"], a["f"]) == (b["iterations"], b["status"], b["hvps"], b["f"]), env
        assert np.array_equal(a["x"], b["x"]), env


@pytest.mark.parametrize("precond", [capi.PRECOND_JACOBI, capi.PRECOND_REGULARIZED_CHOLESKY])
def test_tnt_trial_and_accept_are_the_separate_calls_bit_for_bit(precond):
    """cora_tnt_trial_dev / cora_tnt_accept_dev enqueue what cora_hvp_dev + cora_dots_dev + cora_retract_dev +
    cora_objective_dev / cora_set_point_dev + cora_precondition_projected_dev + cora_dots_dev enqueue, and wait once:
    every scalar and every vector they return is the same bits, with and without the product kept from the trial."""
    P = host.Problem.synthetic(dim=3, n_poses=2500, n_landmarks=4, n_ranges=1200, n_loops=5, seed=5, precond=precond)
    P.update()
    p = 4
    P.set_rank(p)
    P.precond_info()
    dm = P.dims()
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    y, s, hs, xp, pg, hs2, xp2, pg2 = [h.dev_alloc(p) for _ in range(8)]
    rng = np.random.default_rng(3)
    h.upload(rng.uniform(-1, 1, (dm["N"], p)), y)
    h.project_to_manifold_dev(y, y)
    h.set_point_dev(y)
    grad = h.point_ptrs()[2]
    h.upload(1e-2 * rng.uniform(-1, 1, (dm["N"], p)), s)
    h.tangent_space_projection_dev(s, s)
    # the separate calls
    h.hvp_dev(s, hs2)
    o_ref = h.dots_dev([(grad, s), (s, hs2), (s, s)])
    h.retract_dev(s, 1.0, xp2)
    f_ref = h.objective_dev(xp2)
    o = h.tnt_trial_dev(s, hs, xp)
    assert o[:3] == o_ref and o[3] == f_ref
    assert np.array_equal(h.download(hs, p), h.download(hs2, p)) and np.array_equal(h.download(xp, p), h.download(xp2, p))
    # accept: the product kept from the trial ...
    a = h.tnt_accept_dev(xp, pg)
    g_fast = h.download(h.point_ptrs()[2], p)
    e_fast = h.download(h.point_ptrs()[1], p)
    pg_fast = h.download(pg, p)
    # ... against the separate calls at the same point, and against an accept that has no trial to take it from
    h.set_point_dev(xp2)
    grad = h.point_ptrs()[2]
    f2 = h.point_cost()
    h.precondition_projected_dev(grad, pg2)
    n2 = h.dots_dev([(grad, grad), (pg2, pg2), (grad, pg2)])
    assert a == [f2] + n2
    assert np.array_equal(g_fast, h.download(grad, p)) and np.array_equal(e_fast, h.download(h.point_ptrs()[1], p))
    assert np.array_equal(pg_fast, h.download(pg2, p))
    b = h.tnt_accept_dev(xp2, pg)
    assert b == a and np.array_equal(h.download(pg, p), pg_fast)
    for q in (y, s, hs, xp, pg, hs2, xp2, pg2):
        h.dev_free(q)


def test_a_write_to_the_trial_vector_drops_the_kept_product():
    """Round-5 advice: cora_tnt_accept_dev took the product cora_tnt_trial_dev kept whenever the POINTER matched -- after an
    upload into the trial vector, an operation with it as output, or a free / alloc pair that hands the same address out again,
    the accept silently used the Euclidean gradient of the OLD contents.  Every such write now drops the kept product
    (`wrote` in capi.hip; contract in include/cora_hip.h): the accept equals cora_set_point_dev at the new contents."""
    P = host.Problem.synthetic(dim=3, n_poses=1200, n_landmarks=4, n_ranges=600, seed=9, precond=capi.PRECOND_JACOBI)
    P.update()
    p = 4
    P.set_rank(p)
    P.precond_info()
    dm = P.dims()
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    rng = np.random.default_rng(11)
    Y0 = rng.uniform(-1, 1, (dm["N"], p))
    Z = rng.uniform(-1, 1, (dm["N"], p))

    def fresh():
        y, s, hs, xp, pg, z = [h.dev_alloc(p) for _ in range(6)]
        h.upload(Y0, y)
        h.project_to_manifold_dev(y, y)
        h.set_point_dev(y)
        h.upload(1e-2 * Y0[::-1].copy(), s)
        h.tangent_space_projection_dev(s, s)
        h.upload(Z, z)
        h.project_to_manifold_dev(z, z)
        h.tnt_trial_dev(s, hs, xp)
        return y, s, hs, xp, pg, z

    def reference(z):
        h.set_point_dev(z)
        return h.point_cost(), h.download(h.point_ptrs()[1], p)

    # (1) upload into the trial vector
    y, s, hs, xp, pg, z = fresh()
    Zm = h.download(z, p)
    h.upload(Zm, xp)
    a = h.tnt_accept_dev(xp, pg)
    e = h.download(h.point_ptrs()[1], p)
    f_ref, e_ref = reference(z)
    assert a[0] == f_ref and np.array_equal(e, e_ref)
    # (2) the trial vector as the output of an operation
    y, s, hs, xp, pg, z = fresh()
    h.copy_dev(z, p, xp) if hasattr(h, "copy_dev") else h.axpby_dev(1.0, z, 0.0, xp)
    a = h.tnt_accept_dev(xp, pg)
    e = h.download(h.point_ptrs()[1], p)
    f_ref, e_ref = reference(z)
    assert a[0] == f_ref and np.array_equal(e, e_ref)
    # (3) free + alloc: the pool hands the same address out again, zeroed; filled with other contents by a write the
    #     invalidation has already seen
    y, s, hs, xp, pg, z = fresh()
    addr = int(xp) if isinstance(xp, int) else xp
    h.dev_free(xp)
    xp_new = h.dev_alloc(p)
    h.axpby_dev(1.0, z, 0.0, xp_new)
    a = h.tnt_accept_dev(xp_new, pg)
    e = h.download(h.point_ptrs()[1], p)
    f_ref, e_ref = reference(z)
    assert a[0] == f_ref and np.array_equal(e, e_ref), (addr, xp_new)


@pytest.mark.parametrize("d,n,precond", [(3, 30000, capi.PRECOND_REGULARIZED_CHOLESKY), (2, 40000, capi.PRECOND_REGULARIZED_CHOLESKY),
                                          (3, 800, capi.PRECOND_REGULARIZED_CHOLESKY), (3, 800, capi.PRECOND_JACOBI)])
@pytest.mark.parametrize("p", [5, 4])
def test_fused_stpcg_matches_unfused(d, n, precond, p):
    """cora_stpcg_dev folds the residual update with <r, r>, the tangent projection with <r, v> and the step
    with the new direction into three passes, kappa into the product's last block, and -- with a two-stage
    Cholesky solve plan -- all of them into the product and the two sweeps of the solve: <r, v> is taken as |L^-1 r|^2
    from the forward sweep, so that beta is known before the backward sweep and the step / direction update rides on its
    epilogue (five launches per iteration with the solve).  Same iteration as the unfused sequence (CORA_NO_FUSE=1):
    iteration count, M-norm of the step, step, residual and direction agree."""
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=6, n_ranges=n // 2, seed=5, precond=precond)
    P.update()
    P.set_rank(p)
    P.precond_info()
    dm = P.dims()
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    vecs = [h.dev_alloc(p) for _ in range(6)]
    s, r, v, pk, hp, y = vecs
    rng = np.random.default_rng(3)
    h.upload(rng.uniform(-1, 1, (dm["N"], p)), y)
    h.project_to_manifold_dev(y, y)
    h.set_point_dev(y)
    grad = h.point_ptrs()[2]
    out = {}
    env = {"sweep": (), "fused": ("CORA_NO_SWEEP_FUSE", "CORA_NO_INVERSE_FUSE"), "unfused": ("CORA_NO_FUSE",)}
    for mode, names in env.items():
        for var in names:
            os.environ[var] = "1"
        try:
            for delta, iters in ((1e30, 7), (0.5, 40)):   # runs to the limit / stops on the trust-region boundary
                done, step = h.stpcg_dev(grad, delta, s, r, v, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=iters)
                out[(mode, delta)] = (done, step, h.download(s, p), h.download(r, p), h.download(v, p), h.download(pk, p))
            out[mode] = h.stpcg_path()
        finally:
            for var in names:
                os.environ.pop(var, None)
    # the warm start (TNT hands over P g, <g, g>, <g, P g>) runs the same iteration as the cold one
    pg = h.dev_alloc(p)
    h.precondition_projected_dev(grad, pg)
    g_g, g_pg = h.dot_dev(grad, grad, p), h.dot_dev(grad, pg, p)
    done, step = h.stpcg_warm_dev(grad, pg, g_g, g_pg, 0.5, s, r, v, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=40)
    cold = out[("sweep", 0.5)]
    assert done == cold[0] and abs(step - cold[1]) <= 1e-12 * abs(cold[1])
    assert np.abs(h.download(s, p) - cold[2]).max() <= 1e-10 * np.abs(cold[2]).max()
    h.dev_free(pg)
    assert out["unfused"] == 0 and out["fused"] == 1
    # the large plans are two-stage: the sweep-fused form really ran; the small Cholesky plans are one explicit inverse:
    # <r, v> = |W r|^2 and the projection consumes v (five launches); Jacobi: the fused vector passes
    assert out["sweep"] == (2 if n >= 30000 else (3 if precond == capi.PRECOND_REGULARIZED_CHOLESKY else 1))
    for mode in ("sweep", "fused"):
        for delta in (1e30, 0.5):
            a, b = out[(mode, delta)], out[("unfused", delta)]
            assert a[0] == b[0] and a[0] > 0
            assert abs(a[1] - b[1]) <= 1e-10 * abs(b[1])
            # (the sweep-fused form never stores v = P r: the direction update rides on the backward sweep's epilogue,
            # and dV -- a work vector of the C ABI -- is left holding L^-1 r)
            for k in (2, 3, 5) if out[mode] in (2, 3) else (2, 3, 4, 5):
                assert np.abs(a[k] - b[k]).max() <= 1e-9 * np.abs(b[k]).max(), (mode, delta, k)
    for q in vecs:
        h.dev_free(q)
