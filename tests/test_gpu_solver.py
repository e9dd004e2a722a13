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


@pytest.mark.parametrize("d,n,p,loops", [(3, 150, 3, 0), (3, 150, 5, 6), (2, 200, 3, 5)])
def test_tnt_synthetic_noisy(d, n, p, loops):
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=3, n_ranges=n // 2, n_loops=loops, seed=21)
    P.update()
    P.set_rank(p)
    Q, dims = _oracle_problem(P)
    x0 = orc.project_manifold(dims, np.random.default_rng(2).uniform(-1, 1, (dims.N, p)))
    got = P.tnt(x0, max_seconds=120)
    ref = otnt.tnt(Q, dims, x0)
    f0 = orc.cost(Q, x0)
    assert got["f"] < 0.05 * f0  # it optimised (the Jacobi preconditioner is weak on chains)
    # Same algorithm, same landscape: the two runs follow each other step for step.  Converged
    # runs agree to 1e-8; runs stopped by the relative-decrease rule or the iteration limit are
    # compared at 2e-5 (rounding differences accumulate over ~10^4 Hessian-vector products).
    tol = 1e-8 if got["status"] in (0, 1) else 2e-5
    assert abs(got["f"] - ref["f"]) <= tol * abs(ref["f"])
    assert abs(got["iterations"] - (ref["iterations"] - (ref["status"] in ("gradient", "preconditioned_gradient",
                                                                            "iteration_limit")))) <= 2
    assert abs(got["hvps"] - ref["hvps"]) <= 0.02 * ref["hvps"] + 2
    rg = orc.rgrad(Q, dims, got["x"])
    assert abs(np.linalg.norm(rg) - got["grad_norm"]) < 1e-6 * max(1.0, got["grad_norm"])
    assert abs(orc.cost(Q, got["x"]) - got["f"]) < 1e-10 * abs(got["f"])
