"""Pins the CPU oracle (oracle/) against the reference's golden vectors.

Mirrors reference tests/test_optimizer_helpers.cpp:13-53,
tests/test_certification.cpp:81-125, tests/test_construct_problem.cpp and
tests/test_parse_pyfg.cpp, at 1e-9 (the reference itself uses 1e-6)."""
import os

import numpy as np
import pytest

from conftest import EXPECTED_COST, GOLDEN
from mmio import read_dense, read_mm
from oracle import assemble as asm
from oracle import oracle as orc

TOL = 1e-9


def load(case):
    g = asm.parse_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg"))
    A = asm.assemble(g)
    Q = orc.CSR.from_scipy(A["Q"])
    dm = orc.Dims(A["d"], A["n"], A["r"], A["N"])
    return A, Q, dm


def test_assembly_submatrices(case):
    A, Q, dm = load(case)
    for name in ["Arange", "OmegaRange", "RangeDistances", "Apose", "OmegaPose", "T",
                 "RotConLaplacian"]:
        exp = read_mm(os.path.join(GOLDEN, case, name + ".mm"))
        got = A[name]
        if exp.shape == (0, 0):
            assert got.shape[0] == 0 or got.shape[1] == 0, name
            continue
        assert exp.shape == got.shape, name
        assert abs(exp - got).max() < 1e-12, name
    exp = read_mm(os.path.join(GOLDEN, case, "DataMatrix.mm"))
    assert exp.shape == A["Q"].shape
    assert abs(exp - A["Q"]).max() < 1e-12


def test_gt_in_nullspace(case):
    # reference tests/test_construct_problem.cpp: Q * X_gt == 0 (noiseless data)
    A, Q, dm = load(case)
    Xgt = read_dense(os.path.join(GOLDEN, case, "X_gt.mm"))
    # single_rpm / single_range are exact; the RA-SLAM fixture's ranges carry 9
    # decimals (x ~20 m x weights) so its residual is ~1e-8
    tol = 1e-7 if case == "small_ra_slam_problem" else 1e-12
    assert np.abs(orc.spmm(Q, Xgt)).max() < tol


def test_cost_egrad_rgrad_hvp(case):
    A, Q, dm = load(case)
    Y = read_dense(os.path.join(GOLDEN, case, "X_rand_dim2.mm"))
    assert abs(orc.cost(Q, Y) - EXPECTED_COST[case]) < TOL * max(1, abs(EXPECTED_COST[case]))
    eg = orc.egrad(Q, Y)
    assert np.abs(eg - read_dense(os.path.join(GOLDEN, case, "expected_egrad.mm"))).max() < TOL
    assert np.abs(orc.spmm(Q, Y, rowwise=True) - eg).max() == 0.0
    rg = orc.rgrad(Q, dm, Y)
    assert np.abs(rg - read_dense(os.path.join(GOLDEN, case, "expected_rgrad.mm"))).max() < TOL
    dX = read_dense(os.path.join(GOLDEN, case, "rand_dX.mm"))
    hv = orc.hvp(Q, dm, Y, eg, dX)
    assert np.abs(hv - read_dense(os.path.join(GOLDEN, case, "hessProd.mm"))).max() < TOL


def test_lambda_and_certificate(case):
    A, Q, dm = load(case)
    Xgt = read_dense(os.path.join(GOLDEN, case, "X_gt.mm"))
    Lst, lob = orc.lambda_blocks(Q, dm, Xgt)
    assert np.abs(Lst).max(initial=0) < 1e-6 and np.abs(lob).max(initial=0) < 1e-6
    Y = read_dense(os.path.join(GOLDEN, case, "X_rand_dim2.mm"))
    S = orc.certificate_matrix_dense(Q, dm, Y)
    Sexp = read_mm(os.path.join(GOLDEN, case, "S_rand.mm")).toarray()
    assert np.abs(S - Sexp).max() < TOL
    # S-apply operator agrees with the dense certificate matrix
    Lst, lob = orc.lambda_blocks(Q, dm, Y)
    X = np.random.default_rng(0).standard_normal((dm.N, 3))
    assert np.abs(orc.S_apply(Q, dm, Lst, lob, X) - S @ X).max() < 1e-9
    # Hvp == Proj_Y(S Ydot): the identity the fused kernel relies on
    dX = read_dense(os.path.join(GOLDEN, case, "rand_dX.mm"))
    hv = orc.tangent_proj(dm, Y, orc.S_apply(Q, dm, Lst, lob, dX))
    assert np.abs(hv - read_dense(os.path.join(GOLDEN, case, "hessProd.mm"))).max() < TOL
    # certification sign (reference tests/test_certification.cpp:111-124)
    lam_min_gt = np.linalg.eigvalsh(orc.certificate_matrix_dense(Q, dm, Xgt)).min()
    assert lam_min_gt + 1e-6 > 0
    assert np.linalg.eigvalsh(S).min() < -1e-6
    # Cholesky-success criterion (src/CORA_utils.cpp:36-51) agrees with the spectrum
    import scipy.sparse as sp
    Sgt = sp.csr_matrix(orc.certificate_matrix_dense(Q, dm, Xgt) + 1e-6 * np.eye(dm.N))
    assert orc.Cholesky(orc.CSR.from_scipy(Sgt)).ok
    Sr = sp.csr_matrix(S + 1e-6 * np.eye(dm.N))
    assert not orc.Cholesky(orc.CSR.from_scipy(Sr)).ok


def test_manifold_properties(case):
    # reference tests/test_geometry.cpp:11-82 (oblique) + the Stiefel analogue
    A, Q, dm = load(case)
    rng = np.random.default_rng(1)
    for p in (2, 3, 5):
        if p < dm.d:
            continue
        Araw = rng.uniform(-1, 1, (dm.N, p))
        Y = orc.project_manifold(dm, Araw)
        for i in range(dm.n):
            B = Y[i * dm.d:(i + 1) * dm.d]
            assert np.abs(B @ B.T - np.eye(dm.d)).max() < 1e-12
            # polar factor == U V^T of numpy's SVD
            U, s, Vt = np.linalg.svd(Araw[i * dm.d:(i + 1) * dm.d], full_matrices=False)
            assert np.abs(B - U @ Vt).max() < 1e-10
        for j in range(dm.dn, dm.dn + dm.r):
            assert abs(np.linalg.norm(Y[j]) - 1) < 1e-13
        assert np.array_equal(Y[dm.dn + dm.r:], Araw[dm.dn + dm.r:])
        assert np.abs(orc.project_manifold(dm, Y) - Y).max() < 1e-12  # idempotent
        V = orc.tangent_proj(dm, Y, rng.standard_normal((dm.N, p)))
        assert np.abs(orc.tangent_proj(dm, Y, V) - V).max() < 1e-12  # idempotent
        for i in range(dm.n):  # tangent: Y_i V_i^T skew
            M = Y[i * dm.d:(i + 1) * dm.d] @ V[i * dm.d:(i + 1) * dm.d].T
            assert np.abs(M + M.T).max() < 1e-12
        for j in range(dm.dn, dm.dn + dm.r):
            assert abs(Y[j] @ V[j]) < 1e-12
        R = orc.retract(dm, Y, 0.1 * V)
        assert np.abs(R - orc.project_manifold(dm, Y + 0.1 * V)).max() == 0.0


def test_cholesky_vs_dense():
    # reference tests/test.cpp:25-147: block Cholesky solve vs dense inverse
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    for n in (3, 17, 99):
        B = sp.random(n, n, density=0.2, random_state=int(rng.integers(1 << 30)))
        A = (B @ B.T + sp.eye(n) * 0.5).tocsr()
        F = orc.Cholesky(orc.CSR.from_scipy(A))
        assert F.ok
        rhs = rng.standard_normal((n, 4))
        assert np.abs(F.solve(rhs) - np.linalg.solve(A.toarray(), rhs)).max() < 1e-9
        perm = rng.permutation(n).astype(np.int32)
        F2 = orc.Cholesky(orc.CSR.from_scipy(A), perm)
        assert np.abs(F2.solve(rhs) - np.linalg.solve(A.toarray(), rhs)).max() < 1e-9
    # indefinite matrix -> factorisation must fail
    A = sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 1.0]]))
    assert not orc.Cholesky(orc.CSR.from_scipy(A)).ok


def test_precond_semantics(case):
    # blockCholeskySolve (src/CORA_preconditioners.cpp:46-83): N-1 leading rows
    # solved, last row zeroed, then projected (src/CORA.cpp:86-92)
    import scipy.sparse as sp
    A, Q, dm = load(case)
    Qs = Q.to_scipy()
    lam = 1e-3
    M = (Qs + lam * sp.eye(dm.N)).tocsr()[:dm.N - 1, :dm.N - 1]
    F = orc.Cholesky(orc.CSR.from_scipy(M))
    assert F.ok
    Y = read_dense(os.path.join(GOLDEN, case, "X_rand_dim2.mm"))
    V = np.random.default_rng(5).standard_normal(Y.shape)
    out = F.precond(dm, Y, V)
    ref = np.zeros_like(V)
    ref[:dm.N - 1] = np.linalg.solve(M.toarray(), V[:dm.N - 1])
    ref = orc.tangent_proj(dm, Y, ref)
    assert np.abs(out - ref).max() < 1e-9
    jac = orc.precond_jacobi(Q, dm, Y, V)
    assert np.abs(jac - orc.tangent_proj(dm, Y, V / Qs.diagonal()[:, None])).max() < 1e-12


def test_implicit_restatement_is_the_partial_minimum(case):
    """oracle.Implicit (src/CORA_problem.cpp:714-753, 1168-1197): f_impl(Y) equals the explicit cost
    at the analytically recovered translations, and no other translations do better."""
    _, Q, dims = load(case)
    if dims.N - dims.dn - dims.r < 2:
        pytest.skip("needs two translational states")
    I = orc.Implicit(Q, dims)
    rng = np.random.default_rng(0)
    Y = orc.project_manifold(I.dm, rng.uniform(-1, 1, (I.dm.N, dims.d + 1)))
    X = I.translation_explicit(Y)
    f = I.cost(Y)
    assert abs(orc.cost(Q, X) - f) < 1e-9 * max(1.0, abs(f))
    for _ in range(5):
        Xp = X.copy()
        Xp[I.dm.N:-1] += 1e-2 * rng.standard_normal(Xp[I.dm.N:-1].shape)
        assert orc.cost(Q, Xp) >= f - 1e-12 * max(1.0, abs(f))


def test_threaded_numa_placed_oracle_is_the_single_thread_oracle():
    """The all-core column of bench.py runs the oracle's row-parallel loops on bound threads over copies of Q and of the
    vectors that each thread has touched first (oracle.numa_csr / numa_dense / bind_threads).  Every row is computed by
    one thread in the same order, so the Hessian-vector product must equal the single-thread one bit for bit; the
    binding is lifted afterwards (threads other code starts must not inherit a one-core mask)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    n, d, r = 7000, 3, 3000                      # above the sizes where the loops go parallel
    N = n * (d + 1) + r
    nz = 4 * N
    A = sp.coo_matrix((rng.uniform(-1, 1, nz), (rng.integers(0, N, nz), rng.integers(0, N, nz))), shape=(N, N)).tocsr()
    Qs = (A + A.T + sp.eye(N) * 5.0).tocsr()
    Q = orc.CSR.from_scipy(Qs)
    dims = orc.Dims(d, n, r, N)
    Y = orc.project_manifold(dims, rng.uniform(-1, 1, (N, 4)))
    G = orc.egrad(Q, Y)
    V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, (N, 4)))
    orc.set_threads(1)
    ref = orc.hvp(Q, dims, Y, G, V)
    before = os.sched_getaffinity(0)
    try:
        orc.set_threads(2)
        used = orc.bind_threads(2)
        assert len(used) in (0, 2)
        Qn = orc.numa_csr(Q)
        Yn, Gn, Vn = orc.numa_dense(Y), orc.numa_dense(G), orc.numa_dense(V)
        out, work = orc.numa_dense(shape=Y.shape), orc.numa_dense(shape=Y.shape)
        got = orc.hvp(Qn, dims, Yn, Gn, Vn, out=out, work=work)
        assert got is out and np.array_equal(got, ref)
    finally:
        orc.bind_threads(0)
        orc.set_threads(1)
    assert os.sched_getaffinity(0) == before
