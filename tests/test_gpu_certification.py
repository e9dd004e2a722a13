"""Certification on the GPU path, mirroring reference tests/test_certification.cpp:45-125,
plus sign agreement with the oracle (Cholesky success of S + eta I and lambda_min(S))."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN
from cora_amd import capi, host
from mmio import read_dense, read_mm
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _check_certified(res):
    assert res["is_certified"]
    assert abs(res["theta"]) < 1e-6
    assert np.abs(res["x"]).max() < 1e-6


def _check_not_certified(res, theta, x):
    assert not res["is_certified"]
    assert abs(res["theta"] - theta) < 1e-6
    assert min(np.abs(res["x"] - x).max(), np.abs(res["x"] + x).max()) < 1e-6


@pytest.mark.parametrize("n", [10, 1000])
def test_generic_verification(n):
    # reference testIdentityMatrixVerification (n = 10 dense path, n = 1000 LOBPCG path)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, n)
    x /= np.linalg.norm(x)
    I = np.eye(n)
    _check_certified(host.fast_verification(sp.csr_matrix(I), 0.0, X0=x))
    _check_certified(host.fast_verification(sp.csr_matrix(I), 0.0, nx=1))
    A = I - np.outer(x, x)
    _check_certified(host.fast_verification(sp.csr_matrix(A), 1e-8, X0=x))
    _check_certified(host.fast_verification(sp.csr_matrix(A), 1e-8, nx=1))
    B = I - 2 * np.outer(x, x)
    _check_not_certified(host.fast_verification(sp.csr_matrix(B), 0.0, X0=x), -1.0, x)
    _check_not_certified(host.fast_verification(sp.csr_matrix(B), 0.0, nx=1), -1.0, x)


def test_small_ra_slam_verification():
    case = "small_ra_slam_problem"
    P = host.Problem.from_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg"))
    P.update()
    Xgt = read_dense(os.path.join(GOLDEN, case, "X_gt.mm"))
    res = P.certify(Xgt, 1e-6, nx=1)
    _check_certified(res)
    X0 = read_dense(os.path.join(GOLDEN, case, "X_rand_dim2.mm"))
    res = P.certify(X0, 1e-6, nx=1)
    assert not res["is_certified"]
    S = read_mm(os.path.join(GOLDEN, case, "S_rand.mm")).toarray()
    assert abs(res["theta"] - res["x"] @ S @ res["x"]) < 1e-6
    assert abs(res["theta"] - np.linalg.eigvalsh(S)[0]) < 1e-6  # N = 24: dense path gives lambda_min


@pytest.mark.parametrize("d,n,p", [(3, 300, 3), (2, 400, 4)])
def test_certification_sign_matches_oracle(d, n, p):
    """Large-N path (LOBPCG on the device operator): the decision must agree exactly with the
    oracle's Cholesky test, and a returned direction must have curvature < -eta/2."""
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=3, n_ranges=n // 2, seed=17,
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    P.set_rank(p)
    dm = P.dims()
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    rng = np.random.default_rng(4)
    Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dims.N, p)))
    sol = P.tnt(Y, max_seconds=60)
    for point in (Y, sol["x"]):
        f = orc.cost(Q, point)
        eta = min(max(f * 5e-6, 1e-7), 1e-1)  # src/CORA.cpp:112-114,154
        res = P.certify(point, eta)
        Sd = orc.certificate_matrix_dense(Q, dims, point)
        ok = orc.Cholesky(orc.CSR.from_scipy(sp.csr_matrix(Sd + eta * np.eye(dims.N)))).ok
        assert res["is_certified"] == ok
        lam_min = np.linalg.eigvalsh(Sd)[0]
        assert (lam_min + eta > 0) == ok or abs(lam_min + eta) < 1e-9 * abs(lam_min)
        if not ok:
            x = res["x"]
            assert abs(np.linalg.norm(x) - 1) < 1e-8
            assert abs(res["theta"] - x @ Sd @ x) < 1e-8 * max(1.0, abs(res["theta"]))
            assert res["theta"] < -eta / 2
            assert res["theta"] >= lam_min - 1e-8 * abs(lam_min)


def test_ildl_preconditioned_lobpcg_branch():
    """Step 3 of fast_verification (src/CORA_utils.cpp:129-167): a matrix whose negative eigenvalue is tiny and whose
    spectrum is crowded near it -- a long weighted path Laplacian shifted down -- is out of reach of the 1 % of
    unpreconditioned iterations, so the ILDL-preconditioned branch has to find the direction.  Checked against the
    dense eigen-decomposition; the same run without the preconditioner does not get there."""
    n = 4000
    rng = np.random.default_rng(1)
    w = rng.uniform(0.5, 2.0, n - 1)
    Lp = sp.diags([np.r_[w, 0] + np.r_[0, w], -w, -w], [0, 1, -1]).tocsr()
    lam = np.linalg.eigvalsh(Lp.toarray())
    S = (Lp - (lam[1] * 0.5) * sp.eye(n)).tocsr()        # lambda_min(S) = -lam[1]/2 < 0, tiny; second one positive
    lmin = -0.5 * lam[1]
    eta = 1e-3 * abs(lmin)
    x0 = rng.standard_normal((n, 2))
    got = host.fast_verification(S, eta, X0=x0, max_iters=400, lab=dict(seed=False, ildl=True))
    assert not got["is_certified"] and got["step3"]
    assert got["theta"] < -eta / 2                        # the stopping rule of the reference
    assert got["theta"] >= lmin - 1e-12 * max(1.0, abs(lmin))
    assert abs(got["theta"] - got["x"] @ (S @ got["x"])) < 1e-12
    assert 4 <= got["iters"] < 100                        # 1 % unpreconditioned = 4 iterations, then a few more
    # the only direction of negative curvature is the constant vector; the iteration stops as soon as the curvature
    # passes -eta/2 (the reference's rule), so the estimate is dominated by it without having converged to it
    v = np.full(n, 1.0 / np.sqrt(n))
    assert abs(v @ got["x"]) > 0.5
    plain = host.fast_verification(S, eta, X0=x0, max_iters=400, lab=dict(seed=False, ildl=False))
    assert plain["step3"] and plain["iters"] > 2 * got["iters"]
    # with the seed of the failed factorisation (the default) the direction is there at once
    seeded = host.fast_verification(S, eta, X0=x0, max_iters=400, lab=dict())
    assert not seeded["is_certified"] and seeded["theta"] < -eta / 2 and seeded["theta"] >= lmin - 1e-12


@pytest.mark.parametrize("ka,kb", [(10, 10), (1, 1), (3, 17), (16, 16), (17, 5), (24, 24), (20, 10)])
def test_gram_and_combine_blocks(ka, kb):
    """The Rayleigh-Ritz building blocks of LOBPCG (the reference takes the eigensolver from libs/Optimization,
    src/CORA_utils.cpp:113-119): G = A^T B on the fp64 matrix cores (v_mfma_f64_16x16x4, rows as the inner dimension)
    and Out = X C, against numpy on asymmetric random blocks; row count not a multiple of anything."""
    P = host.Problem.synthetic(dim=3, n_poses=2311, n_landmarks=3, n_ranges=1157, seed=4)
    P.update()
    dm = P.dims()
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    rng = np.random.default_rng(ka * 100 + kb)
    A = rng.uniform(-1, 1, (dm["N"], ka)) * np.linspace(0.5, 2.0, ka)
    B = rng.uniform(-1, 1, (dm["N"], kb)) + 0.1 * np.arange(kb)
    a, b, o = h.dev_alloc(ka), h.dev_alloc(kb), h.dev_alloc(kb)
    h.upload(A, a)
    h.upload(B, b)
    G = h.gram_dev(a, ka, b, kb)
    ref = A.T @ B
    assert G.shape == ref.shape
    assert np.abs(G - ref).max() <= 1e-12 * np.abs(A).T.dot(np.abs(B)).max()
    Cm = rng.uniform(-1, 1, (ka, kb))
    h.combine_dev([a], [ka], [Cm], kb, o)
    out = h.download(o, kb)
    assert np.abs(out - A @ Cm).max() <= 1e-13 * np.abs(A @ Cm).max() * ka
    # two input blocks of different widths (the form of the LOBPCG updates: X C1 + W C2)
    C1, C2 = rng.uniform(-1, 1, (ka, kb)), rng.uniform(-1, 1, (kb, kb))
    h.combine_dev([a, b], [ka, kb], [C1, C2], kb, o)
    out = h.download(o, kb)
    ref2 = A @ C1 + B @ C2
    assert np.abs(out - ref2).max() <= 1e-13 * np.abs(ref2).max() * (ka + kb)
    for q in (a, b, o):
        h.dev_free(q)


def test_certificate_matrix_keeps_its_pattern_between_points():
    """Problem::get_certificate_matrix (src/CORA_problem.cpp:1162-1166): S = Q - Lambda(Y).  The host keeps S between
    certifications and rewrites only the entries under Lambda (the d x d pose blocks and the range diagonal); a second
    and a third point, at another rank too, must give the S a fresh merge gives -- checked against the oracle's dense S."""
    P = host.Problem.synthetic(dim=3, n_poses=150, n_landmarks=3, n_ranges=90, n_loops=2, seed=21)
    P.update()
    dm = P.dims()
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    rng = np.random.default_rng(8)
    pattern = None
    for p in (3, 3, 5, 4):
        P.set_rank(p)
        Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dims.N, p)))
        S = P.certificate_matrix(Y)
        ref = orc.certificate_matrix_dense(Q, dims, Y)
        assert np.abs(S.toarray() - ref).max() <= 1e-12 * np.abs(ref).max()
        if pattern is None:
            pattern = (S.indptr.copy(), S.indices.copy())
        assert np.array_equal(S.indptr, pattern[0]) and np.array_equal(S.indices, pattern[1])


@pytest.mark.parametrize("nx,split", [(5, 3), (6, 1), (12, 10)])
def test_start_block_in_pieces_gives_the_same_numbers(nx, split):
    """certify_solution hands the eigensolver its start block as pieces of host memory -- the previous level's
    eigenvectors, cached random columns, the seed of the failed factorisation -- that are uploaded as they are and put side
    by side on the device (LOBPCGSolver::run, src/CORA_problem.cpp:1062-1071 builds the block on the host).  Same block, same
    numbers: theta, the iteration count and the direction must equal the ones of the assembled block bit for bit."""
    n = 3000
    rng = np.random.default_rng(nx)
    w = rng.uniform(0.5, 2.0, n - 1)
    Lp = sp.diags([np.r_[w, 0] + np.r_[0, w], -w, -w], [0, 1, -1]).tocsr()
    S = (Lp + sp.diags(rng.uniform(0.0, 0.3, n)) - 0.2 * sp.eye(n)).tocsr()   # indefinite, not certified at eta = 1e-3
    X0 = rng.uniform(-1, 1, (n, nx))
    whole = host.fast_verification(S, 1e-3, X0=X0, max_iters=300)
    parts = host.fast_verification(S, 1e-3, X0=X0, max_iters=300, split=split)
    assert not whole["is_certified"] and not parts["is_certified"]
    assert whole["theta"] == parts["theta"] and whole["iters"] == parts["iters"]
    assert np.array_equal(whole["x"], parts["x"])


def test_gram_batch_equals_single_products():
    """One Rayleigh-Ritz step of the eigensolver asks for twelve Gram blocks (S_a' A S_b and S_a' S_b, a <= b, over
    X | W | P); cora_gram_batch_dev runs the kernel of cora_gram_dev once per block on pieces of one reduction buffer and
    synchronises once -- the numbers must be those of the single calls bit for bit, whatever the widths."""
    P = host.Problem.synthetic(dim=3, n_poses=3011, n_landmarks=3, n_ranges=1500, seed=9)
    P.update()
    dm = P.dims()
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    rng = np.random.default_rng(12)
    widths = [5, 5, 6, 1, 24, 13]
    blocks = []
    for k in widths:
        q = h.dev_alloc(k)
        h.upload(rng.uniform(-1, 1, (dm["N"], k)) * np.linspace(1.0, 3.0, k), q)
        blocks.append((q, k))
    pairs = [(blocks[i][0], blocks[i][1], blocks[j][0], blocks[j][1]) for i in range(len(blocks)) for j in (0, 2, 4, 5)][:16]
    got = h.gram_batch_dev(pairs)
    for (a, ka, b, kb), G in zip(pairs, got):
        assert np.array_equal(G, h.gram_dev(a, ka, b, kb))
    for q, _ in blocks:
        h.dev_free(q)


def test_start_block_with_columns_of_very_different_size():
    """The eigensolver orthonormalises its start block through the block's Gram matrix.  At 10^6 poses the block holds
    columns of the iterate (translation rows: norm 1e8) next to the unit-length seed of the failed factorisation, and
    the unscaled Gram matrix lost the small column in the rounding of the large ones ("initial block is rank
    deficient" in the middle of a staircase).  Columns are scaled to unit length first; the same situation in small:
    the path Laplacian of test_ildl_preconditioned_lobpcg_branch, one start column blown up by 1e9."""
    n = 4000
    rng = np.random.default_rng(1)
    w = rng.uniform(0.5, 2.0, n - 1)
    Lp = sp.diags([np.r_[w, 0] + np.r_[0, w], -w, -w], [0, 1, -1]).tocsr()
    lam = np.linalg.eigvalsh(Lp.toarray())
    S = (Lp - (lam[1] * 0.5) * sp.eye(n)).tocsr()
    lmin = -0.5 * lam[1]
    eta = 1e-3 * abs(lmin)
    x0 = rng.standard_normal((n, 2))
    x0[:, 0] *= 1e9
    got = host.fast_verification(S, eta, X0=x0, max_iters=400, lab=dict())
    assert not got["is_certified"] and got["theta"] < -eta / 2 and got["theta"] >= lmin - 1e-12
    assert abs(np.linalg.norm(got["x"]) - 1.0) < 1e-9


def test_ritz_block_left_on_the_device_gives_the_numbers_of_the_host_hand_over():
    """solveCORA keeps the eigensolver's Ritz block on the device from one certification to the next
    (Problem::certify_solution_resident: all_eigvecs stays empty, one column comes back, the next start block is picked
    up where it is).  Same decision, theta, iteration counts and direction as handing all_eigvecs over through the host
    (the reference's flow, src/CORA.cpp:158-170), bit for bit."""
    P, gt = host.Problem.synthetic(dim=3, n_poses=4000, n_landmarks=5, n_ranges=2500, seed=12, precond=capi.PRECOND_JACOBI,
                                   ground_truth=True)
    P.update()
    p = 4
    P.set_rank(p)
    Y = P.op("projectToManifold", np.hstack([gt, np.zeros((gt.shape[0], p - gt.shape[1]))])
             + 0.05 * np.random.default_rng(4).standard_normal((gt.shape[0], p)))
    a = P.certify_chain(Y, 1e-3, nx=8, resident=False)
    b = P.certify_chain(Y, 1e-3, nx=8, resident=True)
    assert not a["first"]["is_certified"] and a["first"]["theta"] < -0.5e-3
    for k in ("first", "second"):
        assert a[k]["is_certified"] == b[k]["is_certified"] and a[k]["iters"] == b[k]["iters"]
        assert a[k]["theta"] == b[k]["theta"]
    assert np.array_equal(a["second"]["x"], b["second"]["x"])
