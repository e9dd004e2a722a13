"""CPU checks of the C++ host (cora_amd/csrc/host): PyFG ingestion and data-matrix
assembly against the reference's golden matrices (reference tests/test_parse_pyfg.cpp,
tests/test_construct_problem.cpp) and against the numpy oracle; the synthetic
generator round-trips through its own PyFG output."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from cora_amd import capi, host
from mmio import read_mm
from oracle import assemble as asm

NAMES = ["Arange", "OmegaRange", "RangeDistances", "Apose", "OmegaPose", "T", "RotConLaplacian", "DataMatrix"]


def test_parse_and_assemble_golden(case):
    P = host.Problem.from_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg"))
    P.update()
    for name in NAMES:
        exp = read_mm(os.path.join(GOLDEN, case, name + ".mm"))
        got = P.scipy_matrix(name)
        if exp.shape == (0, 0):
            assert got.shape[0] == 0 or got.shape[1] == 0, name
            continue
        assert exp.shape == got.shape, name
        assert abs(exp - got).max() < 1e-12, name
    dm = P.dims()
    A = asm.assemble(asm.parse_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg")))
    assert (dm["d"], dm["n"], dm["l"], dm["r"], dm["N"]) == (A["d"], A["n"], A["l"], A["r"], A["N"])


@pytest.mark.parametrize("dim,loops", [(2, 0), (3, 7)])
def test_synthetic_roundtrip_and_oracle(tmp_path, dim, loops):
    path = str(tmp_path / "synth.pyfg")
    P = host.Problem.synthetic(dim=dim, n_poses=120, n_landmarks=3, n_ranges=60, n_loops=loops, seed=5,
                               pyfg_out=path)
    P.update()
    Q1 = P.scipy_matrix()
    # the emitted PyFG parses back to the same data matrix (17 significant digits)
    P2 = host.Problem.from_pyfg(path)
    P2.update()
    Q2 = P2.scipy_matrix()
    assert Q1.shape == Q2.shape
    assert abs(Q1 - Q2).max() < 1e-9 * abs(Q1).max()
    # ... and to the oracle's independent parser + assembly
    A = asm.assemble(asm.parse_pyfg(path))
    assert abs(A["Q"] - Q2).max() < 1e-9 * abs(Q2).max()
    dm = P.dims()
    assert dm["n"] == 120 and dm["l"] == 3 and dm["r"] == 60 and dm["rpm"] >= 119
    assert abs(Q1 - Q1.T).max() < 1e-9 * abs(Q1).max()
    # connection-Laplacian structure: rows of Q sum the constant translation null vector
    ones = np.zeros(dm["N"])
    ones[dm["d"] * dm["n"] + dm["r"]:] = 1.0
    assert np.abs(Q1 @ ones).max() < 1e-6 * abs(Q1).max()


def test_registry_errors(tmp_path):
    bad = tmp_path / "dup.pyfg"
    bad.write_text("VERTEX_SE2 0 A0 0 0 0\nVERTEX_SE2 1 A0 1 0 0\n")
    with pytest.raises(host.HostError, match="Pose variable already exists"):
        host.Problem.from_pyfg(str(bad))
    bad.write_text("VERTEX_SE2 0 A0 0 0 0\nFOO 1 2\n")
    with pytest.raises(host.HostError, match="Unknown item type"):
        host.Problem.from_pyfg(str(bad))
    bad.write_text("VERTEX_XY L0 1 2\nVERTEX_XY L1 1 3\nEDGE_RANGE 0 L0 L1 2.0 1.0\nEDGE_RANGE 0 L1 L0 2.0 1.0\n")
    with pytest.raises(host.HostError, match="Range measurement already exists"):
        host.Problem.from_pyfg(str(bad))
    with pytest.raises(host.HostError, match="Could not open file"):
        host.Problem.from_pyfg(str(tmp_path / "missing.pyfg"))
    # operators need the data matrix (checkUpToDate) and then a GPU: no CPU fallback
    P = host.Problem.synthetic(dim=2, n_poses=10, n_landmarks=1, n_ranges=5)
    with pytest.raises(host.HostError, match="data matrix must be constructed"):
        P.op("getRandomInitialGuess")


def test_priors_add_origin_pose(tmp_path):
    f = tmp_path / "prior.pyfg"
    f.write_text("VERTEX_SE2 0 A0 0 0 0\nVERTEX_SE2 1 A1 1 0 0\n"
                 "EDGE_SE2 0 A0 A1 1 0 0 1 0 0 1 0 1\n"
                 "VERTEX_SE2:PRIOR 0 A0 0.5 0.25 0.1 1 0 0 1 0 1\n")
    P = host.Problem.from_pyfg(str(f))
    P.update()
    assert P.dims()["n"] == 3  # A0, A1 and the origin pose O0 (src/CORA_problem.cpp:80-113)
    A = asm.assemble(asm.parse_pyfg(str(f)))
    assert abs(A["Q"] - P.scipy_matrix()).max() < 1e-12


def _submatrices_match(P, case):
    for name in NAMES:
        exp = read_mm(os.path.join(GOLDEN, case, name + ".mm"))
        got = P.scipy_matrix(name)
        if exp.shape == (0, 0):
            assert got.shape[0] == 0 or got.shape[1] == 0, name
            continue
        assert exp.shape == got.shape, name
        assert abs(exp - got).max() < 1e-12, name


def test_construct_single_odometry_programmatically():
    """tests/test_construct_problem.cpp:21-76: two poses, one identity-rotation odometry step of 1 m in x,
    unit covariance, built with the add* methods; the ground truth spans the null space of Q."""
    dim = 2
    P = host.Problem.new(dim, rank=5)
    P.add_pose("x1")
    P.add_pose("x2")
    tran = np.array([1.0, 0.0])
    P.add_rel_pose("x1", "x2", np.eye(2), tran, np.eye(3))
    P.update()
    _submatrices_match(P, "single_rpm")
    Q = P.scipy_matrix()
    rng = np.random.default_rng(0)
    X = np.zeros(((dim + 1) * 2, dim))
    X[0:2] = np.eye(2)
    X[2:4] = np.eye(2)
    X[4] = rng.uniform(-1, 1, dim)
    X[5] = X[4] + tran
    assert np.linalg.norm(Q @ X) < 1e-12
    U, _, _ = np.linalg.svd(rng.standard_normal((dim, dim)))
    assert np.linalg.norm(Q @ (X @ U)) < 1e-12
    with pytest.raises(host.HostError):
        P.add_pose("x1")                      # duplicate variable (src/CORA_problem.cpp:24-31)
    with pytest.raises(host.HostError):
        P.add_rel_pose("x1", "x2", np.eye(2), tran, np.eye(3))   # duplicate measurement (:52-63)


def test_construct_single_range_programmatically():
    """tests/test_construct_problem.cpp:79-131: two landmarks 2 m apart, one range measurement."""
    dim = 3
    P = host.Problem.new(dim, rank=5)
    P.add_landmark("l1")
    P.add_landmark("l2")
    P.add_range("l1", "l2", 2.0, 1.0)
    P.update()
    _submatrices_match(P, "single_range")
    rng = np.random.default_rng(1)
    l1 = rng.uniform(-1, 1, dim)
    u = rng.standard_normal(dim)
    u /= np.linalg.norm(u)
    X = np.vstack([-u, l1, l1 + 2.0 * u])
    assert np.linalg.norm(P.scipy_matrix() @ X) < 1e-12
    with pytest.raises(host.HostError):
        P.add_range("l1", "l3", 1.0, 1.0)     # unknown symbol is only detected at assembly time
        P.update()


def test_trajectory_writers(tmp_path):
    """saveSolnToTum / saveSolnToG20 (src/CORA_utils.cpp:235-346): per-robot pose chains, time stamp = position in
    the chain, quaternion x y z w, 2-D poses lifted to z = 0; invalid rotations are refused (getRotation :211-233)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(4)
    d, n = 3, 4
    P = host.Problem.new(d)
    rots, trans = {}, {}
    for rob in "AB":
        for i in range(n):
            s = "%s%d" % (rob, i)
            P.add_pose(s)
            rots[s], trans[s] = Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix(), rng.uniform(-5, 5, 3)
    for rob in "AB":
        for i in range(n - 1):
            P.add_rel_pose("%s%d" % (rob, i), "%s%d" % (rob, i + 1), np.eye(3), np.ones(3), np.eye(6))
    P.add_range("A0", "B0", 1.0, 1.0)
    P.update()
    dm = P.dims()
    order = ["A%d" % i for i in range(n)] + ["B%d" % i for i in range(n)]   # registration order = pose index
    X = np.zeros((dm["N"], d))
    for k, s in enumerate(order):
        X[k * d:(k + 1) * d] = rots[s].T
        X[d * dm["n"] + dm["r"] + k] = trans[s]
    X[d * dm["n"]] = [1, 0, 0]
    for rob in "AB":
        path = str(tmp_path / ("%s.tum" % rob))
        P.save_trajectory(X, path, robot=rob)
        rows = np.loadtxt(path)
        assert rows.shape == (n, 8) and np.array_equal(rows[:, 0], np.arange(n))
        for i in range(n):
            s = "%s%d" % (rob, i)
            assert np.abs(rows[i, 1:4] - trans[s]).max() < 1e-8
            assert np.abs(Rotation.from_quat(rows[i, 4:8]).as_matrix() - rots[s]).max() < 1e-7
    path = str(tmp_path / "all.g2o")
    P.save_trajectory(X, path, g2o=True)
    lines = open(path).read().splitlines()
    assert len(lines) == 2 * n and all(l.startswith("VERTEX_SE3:QUAT %d " % k) for k, l in enumerate(lines))
    bad = X.copy()
    bad[0:3] *= 1.1
    with pytest.raises(host.HostError, match="determinant|orthogonal"):
        P.save_trajectory(bad, path)
    # 2-D
    P2 = host.Problem.new(2)
    P2.add_pose("A0"); P2.add_pose("A1")
    P2.add_rel_pose("A0", "A1", np.eye(2), np.array([1.0, 0.0]), np.eye(3))
    P2.update()
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    X2 = np.vstack([np.eye(2), R.T, [[0.0, 0.0]], [[1.0, 2.0]]])
    p2 = str(tmp_path / "t.g2o")
    P2.save_trajectory(X2, p2, g2o=True)
    l0, l1 = open(p2).read().splitlines()
    assert l0.split()[:2] == ["VERTEX_SE2", "0"] and abs(float(l1.split()[4]) - th) < 1e-8
    P2.save_trajectory(X2, str(tmp_path / "t.tum"))
    rows = np.loadtxt(str(tmp_path / "t.tum"))
    assert rows[1, 3] == 0.0 and abs(rows[1, 1] - 1.0) < 1e-9 and abs(rows[1, 2] - 2.0) < 1e-9


def test_block_cholesky_free_functions():
    """tests/test.cpp:149-214: 6 x 6 matrix with unit diagonal and 0.5 elsewhere, blocks {3, 3}; the solve is the
    inverse of the block diagonal.  With one more right-hand-side row the last row comes back as zero
    (src/CORA_preconditioners.cpp:78-80)."""
    n = 6
    A = np.full((n, n), 0.5) + 0.5 * np.eye(n)
    Ablk = np.zeros_like(A)
    Ablk[:3, :3], Ablk[3:, 3:] = A[:3, :3], A[3:, 3:]
    inv = np.linalg.inv(Ablk)
    assert np.abs(host.block_cholesky_solve(A, [3, 3], np.eye(n)) - inv).max() < 1e-12
    assert np.abs(host.block_cholesky_solve(A, [3, 3], Ablk) - np.eye(n)).max() < 1e-12
    b = np.arange(1.0, 7.0)
    assert np.abs(host.block_cholesky_solve(A, [3, 3], b)[:, 0] - inv @ b).max() < 1e-12
    # three distinct blocks (the reference's loop would factor the first one three times)
    rng = np.random.default_rng(0)
    M = rng.standard_normal((9, 9))
    S = M @ M.T + 9 * np.eye(9)
    blocks = [2, 3, 4]
    ref = np.zeros((9, 9))
    o = 0
    for s in blocks:
        ref[o:o + s, o:o + s] = np.linalg.inv(S[o:o + s, o:o + s])
        o += s
    assert np.abs(host.block_cholesky_solve(S, blocks, np.eye(9)) - ref).max() < 1e-12
    x = host.block_cholesky_solve(S, blocks, np.vstack([np.eye(9), np.ones((1, 9))]))
    assert x.shape == (10, 9) and np.all(x[-1] == 0.0) and np.abs(x[:9] - ref).max() < 1e-12
    with pytest.raises(host.HostError, match="must sum to A.rows"):
        host.block_cholesky_solve(S, [2, 3], np.eye(9))
    with pytest.raises(host.HostError, match="right-hand side"):
        host.block_cholesky_solve(S, blocks, np.eye(11))


@pytest.mark.parametrize("shift", [1e-3, 50.0, 0.0, -1.0, -200.0])
def test_parallel_cholesky_is_the_sequential_one(shift):
    """The host factorisation (CHOLMOD stand-in) hands disjoint elimination-tree subtrees to threads.  A row's
    arithmetic does not depend on the thread, so the factor, the first failing pivot (the reference's PSD test,
    src/CORA_utils.cpp:36-51) and the direction built from it must be those of the one-thread run, bit for bit."""
    P = host.Problem.synthetic(dim=3, n_poses=6000, n_landmarks=5, n_ranges=3000, n_loops=3, seed=11)
    P.update()
    assert P.dims()["N"] > 20000  # above the size where threads are used
    out = {}
    for threads in ("1", "7"):
        os.environ["CORA_CHOL_THREADS"] = threads
        try:
            out[threads] = P.cholesky_probe(shift=shift)
        finally:
            os.environ.pop("CORA_CHOL_THREADS", None)
    a, b = out["1"], out["7"]
    assert a["ok"] == b["ok"] == (shift > 0.0)  # Q itself is singular (the gauge), so is its leading block
    assert a["nnz"] == b["nnz"] and a["failed_column"] == b["failed_column"]
    assert np.array_equal(a["digest"], b["digest"])
    assert np.array_equal(a["negative_direction"], b["negative_direction"])
    if not a["ok"]:
        z = a["negative_direction"]
        _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
        import scipy.sparse as sp
        N = P.dims()["N"]
        Q = sp.csr_matrix((vals, colidx, rowptr), shape=(N, N))
        assert abs(np.linalg.norm(z) - 1.0) < 1e-12 and z[-1] == 0.0
        assert z @ (Q @ z) + shift * (z @ z) <= 1e-9  # non-positive curvature of (Q + shift I)[0:N-1]


@pytest.mark.parametrize("which,bump", [(0, -1e9), (2, -1e7), (4, -1e12), (5, -1e6), (5, 0.0)])
def test_cholesky_trailing_rows_taken_together(which, bump):
    """The landmark rows close the elimination order and are nearly dense rows of L; choleskyFactor solves them against
    the columns before them on separate threads, forms what they contribute to each other pair by pair and finishes the
    small trailing triangle in order (sparse_cholesky.cpp, steps A-D).  Every number is produced by the sequential
    operations in the sequential order: factor, first failing pivot -- placed here on each of the trailing rows in turn
    by moving one diagonal entry -- and the direction of non-positive curvature must equal the one-thread, one-row-at-a-
    time result bit for bit (the PSD test of the reference, src/CORA_utils.cpp:36-51)."""
    P = host.Problem.synthetic(dim=3, n_poses=6000, n_landmarks=6, n_ranges=3000, n_loops=3, seed=23)
    P.update()
    N = P.dims()["N"]
    assert N > 20000
    row = N - 6 + which
    out = []
    for group_off, threads in (("1", "1"), ("1", "5"), (None, "5"), (None, "16")):
        os.environ["CORA_CHOL_THREADS"] = threads
        if group_off:
            os.environ["CORA_CHOL_NO_TRAILING_GROUP"] = group_off
        try:
            out.append(P.cholesky_probe(m=N, shift=1.0, bump={row: bump}))
        finally:
            os.environ.pop("CORA_CHOL_THREADS", None)
            os.environ.pop("CORA_CHOL_NO_TRAILING_GROUP", None)
    a = out[0]
    assert a["ok"] == (bump == 0.0)
    if not a["ok"]:
        assert a["failed_column"] == N - 6 + which  # the landmarks are the last rows of the order, in index order
    for b in out[1:]:
        assert a["ok"] == b["ok"] and a["nnz"] == b["nnz"] and a["failed_column"] == b["failed_column"]
        assert np.array_equal(a["digest"], b["digest"])
        assert np.array_equal(a["negative_direction"], b["negative_direction"])


def test_solve_plan_takes_the_tree_of_a_complete_factor_from_its_columns(monkeypatch):
    """build_tri_plan (trisolve_build.cpp) recognises a complete Cholesky factor by its pattern being closed under
    elimination and then reads the elimination tree off the first sub-diagonal row of every column instead of running
    Liu's algorithm over the rows of L; CORA_TRI_CHECK_ETREE makes it do both and fail when they differ.  The
    preconditioner's factor (src/CORA_problem.cpp:544-614) of a chain with loop closures and landmarks, above the size
    where the builder's passes run on threads."""
    monkeypatch.setenv("CORA_TRI_CHECK_ETREE", "1")
    P = host.Problem.synthetic(dim=3, n_poses=30000, n_landmarks=6, n_ranges=15000, n_loops=5, seed=4)
    P.update()
    info = P.plan_probe(50.0)
    assert info["stages"] >= 2 and info["nnzL"] > (1 << 20) and info["blocks"] > 16


def test_cholesky_symbolic_cache_changes_nothing():
    """choleskyFactor keeps the pattern-only part of a factorisation (permuted structure, elimination tree, column
    counts) between calls on the same pattern and order -- the certificate matrix S + eta I is factorised several times
    per staircase (src/CORA_utils.cpp:36-51 behind src/CORA_problem.cpp:1030-1103).  Cached and uncached runs must give
    the same factor bit for bit, also when another pattern (another block size m) was factorised in between."""
    P = host.Problem.synthetic(dim=3, n_poses=3000, n_landmarks=4, n_ranges=1500, n_loops=2, seed=5)
    P.update()
    N = P.dims()["N"]
    ref = {}
    os.environ["CORA_CHOL_NO_SYMBOLIC_CACHE"] = "1"
    try:
        for m, shift in ((N - 1, 2.0), (N - 1, 7.5), (N, 3.0), (N - 1, -1.0)):
            ref[(m, shift)] = P.cholesky_probe(m=m, shift=shift)
    finally:
        os.environ.pop("CORA_CHOL_NO_SYMBOLIC_CACHE", None)
    for key in list(ref) + list(ref):  # second round: every pattern is served from the cache
        m, shift = key
        got = P.cholesky_probe(m=m, shift=shift)
        want = ref[key]
        assert got["ok"] == want["ok"] and got["nnz"] == want["nnz"] and got["failed_column"] == want["failed_column"]
        assert np.array_equal(got["digest"], want["digest"])
        assert np.array_equal(got["negative_direction"], want["negative_direction"])
    assert ref[(N - 1, 2.0)]["ok"] and not ref[(N - 1, -1.0)]["ok"]


def test_psd_test_of_the_certificate_agrees_with_the_oracle_and_the_spectrum():
    """The PSD test of fast_verification alone (cora_host_cholesky_test: the host's sparse LL^T in the order certify_solution
    uses, no device) on the certificate matrix of the golden RA-SLAM fixture at the random point: S + shift I has a factor
    exactly when shift > -lambda_min(S) (dense spectrum), and the oracle's Cholesky says the same -- on both sides of the
    threshold and close to it."""
    import ctypes as C
    import scipy.sparse as sp
    from mmio import read_mm
    from oracle import oracle as orc
    case = "small_ra_slam_problem"
    S = sp.csr_matrix(read_mm(os.path.join(GOLDEN, case, "S_rand.mm")))
    S = ((S + S.T) * 0.5).tocsr() if abs(S - S.T).max() > 0 else S.tocsr()
    S.sort_indices()
    P = host.Problem.from_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg"))
    P.update()
    dm = P.dims()
    lam_min = float(np.linalg.eigvalsh(S.toarray())[0])
    assert lam_min < 0
    L = capi.load()
    rp, ci, v = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64)
    for shift in (0.0, -0.5 * lam_min, -lam_min * (1 - 1e-6), -lam_min * (1 + 1e-6), -2 * lam_min, 10.0 - lam_min):
        out = (C.c_int64 * 3)()
        rc = L.cora_host_cholesky_test(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], dm["N"], rp.ctypes.data_as(C.c_void_p),
                                       ci.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.c_double(shift), 2, out)
        assert rc == 0
        Ms = (S + shift * sp.identity(dm["N"])).tocsr()
        Ms.sort_indices()
        expect = shift > -lam_min
        assert bool(out[0]) == expect, (shift, lam_min)
        assert orc.Cholesky(orc.CSR.from_scipy(Ms)).ok == expect, (shift, lam_min)
        assert (out[1] == -1) == expect


def test_oracle_staircase_steps_on_the_golden_fixture():
    """oracle/staircase.py on a case with known answers (noiseless fixture, ground truth X_gt, cost 0): rounding a lifted,
    rotated copy of the ground truth gives the ground truth back up to one rotation; the certification threshold follows
    src/CORA.cpp:111-116; a rank-deficient lift trips certify_solution's singular-value shortcut; the saddle escape from the
    (certified) optimum along any direction finds no decrease and returns the lifted point."""
    from mmio import read_mm
    from oracle import oracle as orc
    from oracle import staircase as ost
    case = "small_ra_slam_problem"
    P = host.Problem.from_pyfg(os.path.join(GOLDEN, case, "factor_graph.pyfg"))
    P.update()
    dm = P.dims()
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    Xgt = np.asfortranarray(read_mm(os.path.join(GOLDEN, case, "X_gt.mm")).toarray())
    assert orc.cost(Q, Xgt) < 1e-9
    rng = np.random.default_rng(4)
    O, _ = np.linalg.qr(rng.standard_normal((4, 4)))
    lifted = np.asfortranarray(np.hstack([Xgt, np.zeros((dm["N"], 2))]) @ O)      # rank 4, same cost
    assert abs(orc.cost(Q, lifted)) < 1e-9
    Yd = ost.project_solution(dims, lifted)
    R = np.linalg.lstsq(Yd, Xgt, rcond=None)[0]
    assert np.abs(R.T @ R - np.eye(dm["d"])).max() < 1e-9 and np.linalg.det(R) > 0 and np.abs(Yd @ R - Xgt).max() < 1e-9
    assert ost.cert_eta(1e-12) == 1e-7 and ost.cert_eta(1e9) == 1e-1 and abs(ost.cert_eta(1000.0) - 5e-3) < 1e-15
    assert ost.rank_deficient(lifted) and not ost.rank_deficient(Xgt)
    v = rng.standard_normal(dm["N"])
    v /= np.linalg.norm(v)
    Yn, info = ost.saddle_escape(Q, dims, lambda Yt, V: orc.tangent_proj(dims, Yt, V), Xgt, -1e-3, v)
    assert not info["accepted"] and not info["fallback"] and np.array_equal(Yn[:, :dm["d"]], Xgt) and np.all(Yn[:, dm["d"]] == 0)
