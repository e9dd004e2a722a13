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
