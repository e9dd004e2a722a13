"""GPU parity of the C++ host (CORA::Problem in cora_amd/csrc/host) against the
oracle: the same checks as reference tests/test_optimizer_helpers.cpp, driven
through the host's own operator methods."""
import os

import numpy as np
import pytest

from conftest import EXPECTED_COST, GOLDEN
from cora_amd import capi, host
from mmio import read_dense
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_host_operators_golden(case):
    P = host.Problem.from_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg"))
    P.update()
    P.set_preconditioner(capi.PRECOND_JACOBI)
    Y = read_dense(os.path.join(GOLDEN, case, "X_rand_dim2.mm"))
    assert abs(P.op("evaluateObjective", Y) - EXPECTED_COST[case]) < 1e-9 * max(1, EXPECTED_COST[case])
    eg = P.op("Euclidean_gradient", Y)
    assert np.abs(eg - read_dense(os.path.join(GOLDEN, case, "expected_egrad.mm"))).max() < 1e-9
    rg = P.op("Riemannian_gradient", Y)
    assert np.abs(rg - read_dense(os.path.join(GOLDEN, case, "expected_rgrad.mm"))).max() < 1e-9
    dX = read_dense(os.path.join(GOLDEN, case, "rand_dX.mm"))
    hv = P.op("Riemannian_Hessian_vector_product", Y, eg, dX)
    assert np.abs(hv - read_dense(os.path.join(GOLDEN, case, "hessProd.mm"))).max() < 1e-9
    st, ob = P.lambda_blocks(read_dense(os.path.join(GOLDEN, case, "X_gt.mm")))
    assert np.abs(st).max(initial=0) < 1e-6 and np.abs(ob).max(initial=0) < 1e-6


def test_host_operators_synthetic_and_shapes():
    P = host.Problem.synthetic(dim=3, n_poses=800, n_landmarks=4, n_ranges=500, n_loops=10, seed=8)
    P.update()
    dm = P.dims()
    P.set_rank(5)
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    Y = P.op("getRandomInitialGuess")
    assert np.abs(Y - orc.project_manifold(dims, Y)).max() < 1e-12  # on the manifold
    rng = np.random.default_rng(1)
    V = P.op("tangent_space_projection", Y, rng.uniform(-1, 1, Y.shape))
    assert np.abs(V - orc.tangent_proj(dims, Y, V)).max() < 1e-12
    R = P.op("retract", Y, 0.2 * V)
    assert np.abs(R - orc.retract(dims, Y, 0.2 * V)).max() < 1e-11
    assert np.abs(P.op("precondition", V) - V / Q.to_scipy().diagonal()[:, None]).max() < 1e-12
    # shape errors surface as the reference's MatrixShapeException text
    with pytest.raises(host.HostError, match="expected matrix of shape"):
        P.L.cora_problem_set_rank(P.h, 4)
        P.op("evaluateObjective", Y)  # Y has 5 columns, rank is now 4
    # the reference's default preconditioner: pinned last row comes back as zero
    P.set_rank(5)
    P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY)
    out = P.op("precondition", V)
    assert np.all(np.isfinite(out)) and np.all(out[-1] == 0.0)


@pytest.mark.parametrize("dim,nsph", [(2, 1), (3, 5)])
def test_oblique_manifold_functions(dim, nsph):
    """tests/test_geometry.cpp:11-88 through the product path: a pose-less problem (landmarks and ranges only)
    is a pure oblique manifold; its operators are the sphere functions the reference tests."""
    P = host.Problem.new(dim, rank=dim, precond=capi.PRECOND_JACOBI)
    for k in range(nsph + 1):
        P.add_landmark("l%d" % k)
    for k in range(nsph):
        P.add_range("l%d" % k, "l%d" % (k + 1), 1.0 + 0.5 * k, 1.0)
    P.update()
    dm = P.dims()
    assert (dm["n"], dm["r"], dm["l"]) == (0, nsph, nsph + 1)
    rng = np.random.default_rng(5)
    Y = P.op("projectToManifold", rng.uniform(-1, 1, (dm["N"], dim)))
    assert np.abs(np.linalg.norm(Y[:nsph], axis=1) - 1).max() < 1e-14         # unit rows
    assert np.abs(P.op("tangent_space_projection", Y, Y)[:nsph]).max() < 1e-14  # Y projects to zero on its own spheres
    V = P.op("tangent_space_projection", Y, rng.uniform(-1, 1, Y.shape))
    assert np.linalg.norm(V[:nsph]) > 1e-6
    assert np.abs(np.sum(V[:nsph] * Y[:nsph], axis=1)).max() < 1e-14          # tangent: <v_j, y_j> = 0
    R = P.op("retract", Y, V)
    assert np.abs(np.linalg.norm(R[:nsph], axis=1) - 1).max() < 1e-14
    assert np.abs(R - Y).max() > 1e-6
    assert np.abs(R[nsph:] - (Y + V)[nsph:]).max() < 1e-15                     # translations are Euclidean


def _rot3(rng, s):
    w = rng.normal(0, s, 3)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def test_mixed_measurement_types_match_oracle():
    """Every measurement type of the reference in one graph, built with the add* methods: two robots, inter-robot
    (pose-pose) ranges, a landmark-landmark range, pose-landmark relative positions, a pose prior and a landmark
    prior (which add the origin pose, src/CORA_problem.cpp:80-113).  Operators against the oracle on the host's
    Q, then the full solve."""
    rng = np.random.default_rng(21)
    d, n = 3, 60
    P = host.Problem.new(d, rank=d, precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    truth = {}
    for rob in "AB":
        R, t = np.eye(3), rng.uniform(-5, 5, 3)
        for i in range(n):
            P.add_pose("%s%d" % (rob, i))
            truth["%s%d" % (rob, i)] = (R.copy(), t.copy())
            dR, dt = _rot3(rng, 0.1), np.array([1.0, 0, 0]) + rng.normal(0, 0.1, 3)
            R, t = R @ dR, t + R @ dt
    for k in range(3):
        P.add_landmark("L%d" % k)
        truth["L%d" % k] = (None, rng.uniform(-20, 20, 3))
    cov = np.diag([0.05 ** 2] * 3 + [0.01 ** 2] * 3)
    for rob in "AB":
        for i in range(n - 1):
            (Ri, ti), (Rj, tj) = truth["%s%d" % (rob, i)], truth["%s%d" % (rob, i + 1)]
            P.add_rel_pose("%s%d" % (rob, i), "%s%d" % (rob, i + 1), Ri.T @ Rj @ _rot3(rng, 0.01),
                           Ri.T @ (tj - ti) + rng.normal(0, 0.05, 3), cov)
    dist = lambda a, b: float(np.linalg.norm(truth[a][1] - truth[b][1]))
    for i in range(0, n, 4):                                   # inter-robot ranges: both ends are poses
        P.add_range("A%d" % i, "B%d" % ((i * 7) % n), dist("A%d" % i, "B%d" % ((i * 7) % n)) + rng.normal(0, 0.1), 0.01)
    for i in range(0, n, 5):                                   # pose -> landmark ranges
        P.add_range("B%d" % i, "L%d" % (i % 3), dist("B%d" % i, "L%d" % (i % 3)) + rng.normal(0, 0.1), 0.01)
    P.add_range("L0", "L1", dist("L0", "L1"), 0.01)            # no pose at either end
    for i in (3, 17, 40):                                      # relative position of a landmark seen from a pose
        Ri, ti = truth["A%d" % i]
        P.add_rel_pose_landmark("A%d" % i, "L2", Ri.T @ (truth["L2"][1] - ti) + rng.normal(0, 0.05, 3), np.eye(3) * 0.05 ** 2)
    P.add_pose_prior("A0", truth["A0"][0], truth["A0"][1], cov)
    P.add_landmark_prior("L0", truth["L0"][1], np.eye(3) * 0.1 ** 2)
    P.update()
    dm = P.dims()
    assert dm["n"] == 2 * n + 1 and dm["l"] == 3               # + the origin pose the priors hang off
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    p = 5
    P.set_rank(p)
    Y = P.op("getRandomInitialGuess")
    assert np.abs(Y - orc.project_manifold(dims, Y)).max() < 1e-12
    f = orc.cost(Q, Y)
    assert abs(P.op("evaluateObjective", Y) - f) < 1e-11 * abs(f)
    G = orc.egrad(Q, Y)
    assert np.abs(P.op("Euclidean_gradient", Y) - G).max() < 1e-11 * np.abs(G).max()
    V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, Y.shape))
    ref = orc.hvp(Q, dims, Y, G, V)
    assert np.abs(P.op("Riemannian_Hessian_vector_product", Y, G, V) - ref).max() < 1e-11 * np.abs(ref).max()
    import scipy.sparse as sp
    lam = P.precond_info()["lam"]
    M = (Q.to_scipy() + lam * sp.eye(dims.N)).tocsr()[:dims.N - 1, :dims.N - 1]
    out = P.op("precondition", V)
    assert np.abs(M @ out[:-1] - V[:-1]).max() < 1e-8 * np.abs(V).max()
    P.set_rank(d)
    res = P.solve(P.op("getRandomInitialGuess"), max_rank=8, max_seconds=60)
    X = res["x"]
    assert abs(orc.cost(Q, X) - res["f"]) < 1e-8 * max(1.0, res["f"])
    nres = 6 * 2 * (n - 1) + dm["r"] + 9 + 6 + 3               # whitened residual count: optimum ~ half the dof
    assert res["f"] < nres
