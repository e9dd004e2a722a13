"""Translation-implicit formulation (Formulation::Implicit, src/CORA_problem.cpp:714-753) on the
GPU path against the oracle's Schur-complement restatement.  The reference ships no test for this
mode (its examples/config.json selects it), so parity is anchored on the identity with the explicit
problem: f_impl(Y) = min_t f_expl([Y; t])."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from cora_amd import capi, host
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _problem(implicit, **kw):
    P = host.Problem.synthetic(**kw)
    P.update()
    P.set_formulation(implicit)
    return P


def _oracle(P):
    dm = P.dims()
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    return Q, dims, orc.Implicit(Q, dims)


@pytest.mark.parametrize("d,n,l,r,p", [(3, 300, 4, 200, 5), (2, 500, 3, 400, 4), (3, 64, 0, 0, 3)])
def test_implicit_operators_match_oracle(d, n, l, r, p):
    P = _problem(True, dim=d, n_poses=n, n_landmarks=l, n_ranges=r, n_loops=10, seed=4, precond=capi.PRECOND_JACOBI)
    Q, dims, I = _oracle(P)
    P.set_rank(p)
    m = d * n + r
    assert P.variable_size() == m
    Y = P.op("getRandomInitialGuess")
    assert Y.shape == (m, p)
    assert np.abs(Y - orc.project_manifold(I.dm, Y)).max() < 1e-12
    f = P.op("evaluateObjective", Y)
    assert abs(f - I.cost(Y)) < 1e-10 * max(1.0, abs(f))
    G = P.op("Euclidean_gradient", Y)
    Gref = I.product(Y)
    scale = np.abs(Gref).max()
    assert np.abs(G - Gref).max() < 1e-10 * scale
    assert np.abs(P.op("Riemannian_gradient", Y) - I.rgrad(Y)).max() < 1e-10 * scale
    rng = np.random.default_rng(3)
    V = P.op("tangent_space_projection", Y, rng.uniform(-1, 1, Y.shape))
    H = P.op("Riemannian_Hessian_vector_product", Y, G, V)
    assert np.abs(H - I.hvp(Y, V)).max() < 1e-10 * scale
    st, ob = P.lambda_blocks(Y)
    st_ref, ob_ref = I.lambda_blocks(Y)
    assert np.abs(st - st_ref).max(initial=0) < 1e-10 * scale and np.abs(ob - ob_ref).max(initial=0) < 1e-10 * scale
    R = P.op("retract", Y, 0.3 * V)
    assert np.abs(R - orc.retract(I.dm, Y, 0.3 * V)).max() < 1e-11
    # translation recovery and the identity with the explicit problem
    X = P.op("getTranslationExplicitSolution", Y)
    assert X.shape == (dims.N, p)
    Xref = I.translation_explicit(Y)
    assert np.abs(X - Xref).max() < 1e-9 * max(1.0, np.abs(Xref).max())
    assert abs(orc.cost(Q, X) - f) < 1e-9 * max(1.0, abs(f))
    # explicit gradient at the lifted point vanishes on the free translations
    gt = orc.egrad(Q, X)[m:-1]
    assert np.abs(gt).max(initial=0) < 1e-8 * scale


def test_implicit_preconditioners():
    P = _problem(True, dim=3, n_poses=400, n_landmarks=3, n_ranges=300, n_loops=5, seed=5, precond=capi.PRECOND_JACOBI)
    Q, dims, I = _oracle(P)
    P.set_rank(4)
    m = I.dm.N
    Y = P.op("getRandomInitialGuess")
    V = P.op("tangent_space_projection", Y, np.random.default_rng(1).uniform(-1, 1, Y.shape))
    # Jacobi: leading block of diag(Q)^-1
    assert np.abs(P.op("precondition", V) - V / Q.to_scipy().diagonal()[:m, None]).max() < 1e-12
    # Cholesky preconditioners: lift with zero translations, solve, take the head (:878-884)
    P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY)
    out = P.op("precondition", V)
    lam = P.precond_info()["lam"]
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    A = (Q.to_scipy() + lam * sp.identity(dims.N)).tocsc()[:dims.N - 1, :dims.N - 1]
    Vl = np.vstack([V, np.zeros((dims.N - m, V.shape[1]))])
    ref = spl.splu(A).solve(Vl[:-1])[:m]
    assert np.abs(out - ref).max() < 1e-8 * np.abs(ref).max()


def test_implicit_tnt_matches_explicit_optimum():
    kw = dict(dim=3, n_poses=150, n_landmarks=3, n_ranges=120, n_loops=6, seed=11,
              precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    Pe, Pi = _problem(False, **kw), _problem(True, **kw)
    Q, dims, I = _oracle(Pi)
    for P in (Pe, Pi):
        P.set_rank(5)
    m = I.dm.N
    x0 = Pe.op("getRandomInitialGuess")
    re = Pe.tnt(x0, grad_tol=1e-7, pgrad_tol=1e-7)
    ri = Pi.tnt(np.asfortranarray(x0[:m]), grad_tol=1e-7, pgrad_tol=1e-7)
    assert ri["x"].shape == (m, 5)
    assert abs(I.cost(ri["x"]) - ri["f"]) < 1e-9 * max(1.0, ri["f"])
    gn = np.linalg.norm(I.rgrad(ri["x"]))
    assert abs(gn - ri["grad_norm"]) < 1e-6 * max(1.0, gn)   # the reported gradient is the implicit one
    # TNT stops on relative decrease (1e-6) in both modes; same problem => same optimum value
    assert ri["f"] < 1e-4 * I.cost(x0[:m])
    assert abs(ri["f"] - re["f"]) < 1e-3 * re["f"]


@pytest.mark.parametrize("case", ["small_ra_slam_problem", "single_range", "single_rpm"])
def test_implicit_solve_fixtures(case):
    P = host.Problem.from_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg"))
    P.update()
    P.set_formulation(True)
    Q, dims, I = _oracle(P)
    x0 = P.op("getRandomInitialGuess")
    res = P.solve(x0, max_rank=10)
    X = res["x"]
    assert X.shape == (I.dm.N, dims.d)
    assert np.abs(X - orc.project_manifold(I.dm, X)).max() < 1e-9
    assert abs(I.cost(X) - res["f"]) < 1e-9 * max(1.0, abs(res["f"]))
    assert res["f"] < 1e-6 and res["certified"]
    full = P.op("alignEstimateToOrigin", X)
    assert full.shape == (dims.N, dims.d)
    if dims.n > 0:
        assert np.abs(full[:dims.d] - np.eye(dims.d)).max() < 1e-9
    assert orc.cost(Q, full) < 1e-6


def test_implicit_solve_synthetic_matches_explicit_cost():
    kw = dict(dim=3, n_poses=120, n_landmarks=4, n_ranges=120, seed=23, precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    Pe, Pi = _problem(False, **kw), _problem(True, **kw)
    Q, dims, I = _oracle(Pi)
    x0 = Pe.op("getRandomInitialGuess")
    re = Pe.solve(x0, max_rank=8, max_seconds=60)
    ri = Pi.solve(np.asfortranarray(x0[:I.dm.N]), max_rank=8, max_seconds=60)
    assert ri["certified"] == re["certified"]
    if re["certified"]:  # both certified => both at the global optimum of the same problem
        assert abs(ri["f"] - re["f"]) < 1e-5 * max(1.0, re["f"])
