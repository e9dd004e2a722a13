"""GPU parity: every C-ABI operator of libcora_hip.so against the CPU oracle and
the reference's golden vectors.  Mirrors reference tests/test_optimizer_helpers.cpp,
tests/test_certification.cpp:81-125 and tests/test_geometry.cpp.

Tolerances (fp64): 1e-9 absolute on the golden fixtures (the reference uses 1e-6);
1e-10 relative (to the largest entry of the result) against the oracle on
synthetic graphs -- only the summation order differs."""
import os

import numpy as np
import pytest

from conftest import EXPECTED_COST, GOLDEN
from cora_amd import capi
from mmio import read_dense, read_mm
from oracle import oracle as orc
from synth import make_problem
from test_oracle_golden import load

pytestmark = pytest.mark.gpu
TOL = 1e-9
REL = 1e-10


def ctx_for(Q, dm, p=None, **kw):
    c = capi.Context(dm.d, dm.n, dm.r, dm.n_trans, Q.rowptr, Q.col, Q.val, **kw)
    if p:
        c.set_rank(p)
    return c


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_golden_cost_gradients_hvp(case):
    A, Q, dm = load(case)
    c = ctx_for(Q, dm, 2)
    Y = read_dense(os.path.join(GOLDEN, case, "X_rand_dim2.mm"))
    assert abs(c.evaluateObjective(Y) - EXPECTED_COST[case]) < TOL * max(1, EXPECTED_COST[case])
    eg = c.Euclidean_gradient(Y)
    assert np.abs(eg - read_dense(os.path.join(GOLDEN, case, "expected_egrad.mm"))).max() < TOL
    assert np.abs(c.dataMatrixProduct(Y) - eg).max() < 1e-12
    rg = c.Riemannian_gradient(Y)
    assert np.abs(rg - read_dense(os.path.join(GOLDEN, case, "expected_rgrad.mm"))).max() < TOL
    dX = read_dense(os.path.join(GOLDEN, case, "rand_dX.mm"))
    hv = c.Riemannian_Hessian_vector_product(Y, eg, dX)
    assert np.abs(hv - read_dense(os.path.join(GOLDEN, case, "hessProd.mm"))).max() < TOL


def test_golden_lambda_and_certificate(case):
    A, Q, dm = load(case)
    c = ctx_for(Q, dm, 2)
    Xgt = read_dense(os.path.join(GOLDEN, case, "X_gt.mm"))
    st, ob = c.compute_Lambda_blocks(Xgt)
    assert np.abs(st).max(initial=0) < 1e-6 and np.abs(ob).max(initial=0) < 1e-6
    Y = read_dense(os.path.join(GOLDEN, case, "X_rand_dim2.mm"))
    st, ob = c.compute_Lambda_blocks(Y)
    st_o, ob_o = orc.lambda_blocks(Q, dm, Y)
    assert np.abs(st - st_o).max(initial=0) < TOL and np.abs(ob - ob_o).max(initial=0) < TOL
    # S = Q - Lambda applied to the identity reproduces S_rand.mm
    S = c.certificate_product(np.eye(dm.N)[:, :min(dm.N, 24)])
    Sexp = read_mm(os.path.join(GOLDEN, case, "S_rand.mm")).toarray()
    assert np.abs(S - Sexp[:, :S.shape[1]]).max() < TOL


def test_manifold_ops_fixture(case):
    A, Q, dm = load(case)
    rng = np.random.default_rng(11)
    for p in (2, 3, 5, 8):
        if p < dm.d:
            continue
        c = ctx_for(Q, dm, p)
        Araw = rng.uniform(-1, 1, (dm.N, p))
        Y = c.projectToManifold(Araw)
        assert np.abs(Y - orc.project_manifold(dm, Araw)).max() < 1e-12
        V = rng.standard_normal((dm.N, p))
        PV = c.tangent_space_projection(Y, V)
        assert np.abs(PV - orc.tangent_proj(dm, Y, V)).max() < 1e-12
        R = c.retract(Y, 0.3 * PV)
        assert np.abs(R - orc.retract(dm, Y, 0.3 * PV)).max() < 1e-12
        c.precond_setup(capi.PRECOND_JACOBI)
        assert np.abs(c.precondition(V) - V / Q.to_scipy().diagonal()[:, None]).max() < 1e-12
        assert abs(c.inner_product(V, PV) - orc.inner(V, PV)) < 1e-10 * abs(orc.inner(V, PV))


@pytest.mark.parametrize("d,n,lm,nr,loops,p", [
    (3, 1500, 4, 800, 0, 5),     # chain, long landmark rows (multi-chunk path)
    (3, 900, 3, 500, 60, 3),     # loop closures
    (2, 1200, 2, 700, 20, 5),    # SE(2)
    (3, 600, 0, 0, 0, 4),        # pose-graph only: no ranges / landmarks
    (3, 700, 3, 400, 0, 10),     # certification block width
    (2, 300, 1, 299, 0, 7),
])
def test_synthetic_operators(d, n, lm, nr, loops, p):
    A, Q, dm = make_problem(d=d, n=n, n_landmarks=lm, n_ranges=nr, n_loops=loops, seed=5)
    c = ctx_for(Q, dm, p)
    rng = np.random.default_rng(2)
    Y = orc.project_manifold(dm, rng.uniform(-1, 1, (dm.N, p)))
    V = orc.tangent_proj(dm, Y, rng.uniform(-1, 1, (dm.N, p)))
    assert relerr(c.dataMatrixProduct(V), orc.spmm(Q, V)) < REL
    f = orc.cost(Q, Y)
    assert abs(c.evaluateObjective(Y) - f) < 1e-11 * abs(f)
    G = orc.egrad(Q, Y)
    assert relerr(c.Euclidean_gradient(Y), G) < REL
    assert relerr(c.Riemannian_gradient(Y), orc.tangent_proj(dm, Y, G)) < REL
    hv_o = orc.hvp(Q, dm, Y, G, V)
    assert relerr(c.Riemannian_Hessian_vector_product(Y, G, V), hv_o) < REL
    # resident path: set_point + hvp_dev (Lambda from the device's own gradient)
    c.set_point(Y)
    assert abs(c.point_cost() - f) < 1e-11 * abs(f)
    x, o = c.dev_alloc(p), c.dev_alloc(p)
    c.upload(V, x)
    c.hvp_dev(x, o)
    assert relerr(c.download(o, p), hv_o) < REL
    # certificate operator at k = 10 columns
    st, ob = orc.lambda_blocks(Q, dm, Y)
    X = rng.standard_normal((dm.N, 10))
    assert relerr(c.certificate_product(X), orc.S_apply(Q, dm, st, ob, X)) < REL
    # tangent vectors stay tangent; Hvp is self-adjoint on the tangent space
    W = orc.tangent_proj(dm, Y, rng.uniform(-1, 1, (dm.N, p)))
    hw = c.Riemannian_Hessian_vector_product(Y, G, W)
    assert abs(orc.inner(V, hw) - orc.inner(W, hv_o)) < 1e-9 * abs(orc.inner(W, hv_o))
    # retraction on device
    c.retract_dev(x, 0.5, o)
    assert np.abs(c.download(o, p) - orc.retract(dm, Y, 0.5 * V)).max() < 1e-11
    # Jacobi precon closure = Proj(D^-1 V)
    c.precond_setup(capi.PRECOND_JACOBI)
    c.precondition_projected_dev(x, o)
    assert relerr(c.download(o, p), orc.precond_jacobi(Q, dm, Y, V)) < REL
    # vector ops
    c.axpby_dev(2.0, x, -0.5, o)
    assert relerr(c.download(o, p), 2.0 * V - 0.5 * orc.precond_jacobi(Q, dm, Y, V)) < 1e-12
    assert abs(c.dot_dev(x, x, p) - orc.inner(V, V)) < 1e-12 * orc.inner(V, V)


def test_rank_change_and_errors():
    A, Q, dm = make_problem(d=3, n=300, n_landmarks=2, n_ranges=150, seed=9)
    c = ctx_for(Q, dm, 3)
    rng = np.random.default_rng(4)
    for p in (3, 4, 5, 6):  # staircase: incrementRank
        c.set_rank(p)
        Y = orc.project_manifold(dm, rng.uniform(-1, 1, (dm.N, p)))
        with pytest.raises(capi.CoraError):  # point invalidated by the rank change
            c.point_cost()
        assert abs(c.evaluateObjective(Y) - orc.cost(Q, Y)) < 1e-11 * orc.cost(Q, Y)
    with pytest.raises(capi.CoraError) as e:
        Ybad = np.zeros((dm.N - 1, 6), order="F")
        c.evaluateObjective(Ybad)  # leading dimension < N -> shape error
    assert e.value.code == 1
    with pytest.raises(capi.CoraError):  # no factor installed yet
        c.precond_setup(capi.PRECOND_REGULARIZED_CHOLESKY)


def test_partitioned_handles_match_single():
    """Row-partitioned handles (the multi-GPU layout) on one device WITHOUT communication (the caller keeps the remote
    rows current): each rank's Hvp shard equals the single-handle result.  Such handles keep their long (landmark) rows
    whole (CORA_PART_WHOLE_LONG_ROWS); the default distributes them and needs the all-reduce of the communication
    (tests/test_gpu_sharded.py)."""
    A, Q, dm = make_problem(d=3, n=800, n_landmarks=4, n_ranges=500, n_loops=10, seed=6)
    p = 5
    rng = np.random.default_rng(8)
    Y = orc.project_manifold(dm, rng.uniform(-1, 1, (dm.N, p)))
    V = orc.tangent_proj(dm, Y, rng.uniform(-1, 1, (dm.N, p)))
    ref = orc.hvp(Q, dm, Y, orc.egrad(Q, Y), V)
    world = 4
    total = np.zeros_like(ref)
    fsum = 0.0
    for rank in range(world):
        c = ctx_for(Q, dm, p, rank=rank, world=world, whole_long_rows=True)
        c.set_point(Y)
        fsum += c.point_cost()
        x, o = c.dev_alloc(p), c.dev_alloc(p)
        c.upload(V, x)
        c.hvp_dev(x, o)
        got = c.download(o, p)
        m = c.row_map()
        mine = (m >= c.shard_begin) & (m < c.shard_begin + c.shard_rows)
        total[mine] = got[mine]
        # only the rows cora_remote_rows lists are read outside the shard: poison all others
        need = c.remote_rows()
        assert need.size and np.all((need < c.shard_begin) | (need >= c.shard_begin + c.shard_rows))
        keep = mine.copy()
        keep |= np.isin(m, need)
        Vp = V.copy()
        Vp[~keep] = np.nan
        c.upload(Vp, x)
        c.hvp_dev(x, o)
        again = c.download(o, p)
        assert np.array_equal(again[mine], got[mine])
        c.close()
    assert relerr(total, ref) < REL
    assert abs(fsum - orc.cost(Q, Y)) < 1e-11 * orc.cost(Q, Y)


@pytest.mark.parametrize("k", [1, 2, 9, 11, 12, 13, 14, 15, 16, 17, 19, 20, 22, 23, 24])
def test_wide_blocks_every_row_stride(k):
    """Every row stride is compiled unpadded (LD = k for 2 <= k <= 24; rounds 1-2 padded 13..24 to 16 / 20 / 24): Q*X
    and (Q - Lambda) X on k-column blocks (the LOBPCG block of certification has max(10, p + 2) columns,
    src/CORA_problem.cpp:1062-1063), plus Gram / combine kernels."""
    A, Q, dm = make_problem(d=3, n=400, n_landmarks=3, n_ranges=250, n_loops=5, seed=12)
    p = 4
    c = ctx_for(Q, dm, p)
    rng = np.random.default_rng(k)
    Y = orc.project_manifold(dm, rng.uniform(-1, 1, (dm.N, p)))
    c.set_point(Y)
    X = rng.standard_normal((dm.N, k))
    assert relerr(c.dataMatrixProduct(X), orc.spmm(Q, X)) < REL
    st, ob = orc.lambda_blocks(Q, dm, Y)
    assert relerr(c.certificate_product(X), orc.S_apply(Q, dm, st, ob, X)) < REL
    assert abs(c.inner_product(X, X) - orc.inner(X, X)) < 1e-12 * orc.inner(X, X)


def test_rank_up_to_24():
    A, Q, dm = make_problem(d=3, n=200, n_landmarks=2, n_ranges=100, seed=3)
    rng = np.random.default_rng(0)
    for p in (9, 11, 13, 16, 18, 21, 24):
        c = ctx_for(Q, dm, p)
        Y = orc.project_manifold(dm, rng.uniform(-1, 1, (dm.N, p)))
        G = orc.egrad(Q, Y)
        V = orc.tangent_proj(dm, Y, rng.uniform(-1, 1, (dm.N, p)))
        assert relerr(c.Riemannian_Hessian_vector_product(Y, G, V), orc.hvp(Q, dm, Y, G, V)) < REL
        assert np.abs(c.retract(Y, 0.3 * V) - orc.retract(dm, Y, 0.3 * V)).max() < 1e-11
        c.close()
    c = ctx_for(Q, dm, 3)
    with pytest.raises(capi.CoraError):
        c.set_rank(25)


@pytest.fixture
def window_form():
    """Forces the LDS-window form of the pose slices (default: from 2 048 slices on) for the duration of a test."""
    L = capi.load()
    old = L.cora_debug_spmm_window_min_slices(0)
    yield
    L.cora_debug_spmm_window_min_slices(old)


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("p", list(range(2, 25)))
def test_window_form_of_every_row_stride(d, p, window_form):
    """The chain slices have two forms (X through LDS windows filled by LDS-DMA with staged stores / direct gathers) and,
    by row stride, three ways to hold their work (translation row fused into the slots, translation row first, one
    wavefront per SIMD).  Small problems take the gather form, so every stride's WINDOW form is run here on a small
    problem -- Q X, (Q - Lambda) X, the Hvp, the Hvp with kappa (through one STPCG iteration's product) -- against the
    oracle; ragged last slice (n not a multiple of 64), loop closures (general slots), several ranges per pose (tails)."""
    if p < d:
        pytest.skip("p >= d")
    A, Q, dm = make_problem(d=d, n=333, n_landmarks=3, n_ranges=500, n_loops=7, seed=40 + p)
    c = ctx_for(Q, dm, p)
    rng = np.random.default_rng(p)
    Y = orc.project_manifold(dm, rng.uniform(-1, 1, (dm.N, p)))
    G = orc.egrad(Q, Y)
    V = orc.tangent_proj(dm, Y, rng.uniform(-1, 1, (dm.N, p)))
    X = rng.standard_normal((dm.N, p))
    c.set_point(Y)
    assert relerr(c.dataMatrixProduct(X), orc.spmm(Q, X)) < REL
    st, ob = orc.lambda_blocks(Q, dm, Y)
    assert relerr(c.certificate_product(X), orc.S_apply(Q, dm, st, ob, X)) < REL
    ref = orc.hvp(Q, dm, Y, G, V)
    assert relerr(c.Riemannian_Hessian_vector_product(Y, G, V), ref) < REL
    if p <= 12:  # the product with the kappa slots: <V, H V> as the device STPCG's first iteration sees it
        y, g, s, r, v, pk, hp = (c.dev_alloc(p) for _ in range(7))
        c.upload(Y, y)
        c.set_point_dev(y)
        c.upload(V, g)
        c.precond_setup(capi.PRECOND_NONE)
        c.stpcg_dev(g, 1e-30, s, r, v, pk, hp, max_iters=1)   # radius ~ 0: one product, then the boundary step
        Hp = c.download(hp, p)
        Pk = c.download(pk, p)
        assert relerr(Hp, orc.hvp(Q, dm, Y, G, Pk)) < REL
    c.close()


@pytest.mark.parametrize("p", [4, 5, 8])
def test_heavy_tails_and_window_form(p, window_form):
    """More than 64 pairs in a slice's tail (three range measurements per pose on average: the second and third round
    of the wavefront's gather), in both forms of the kernel."""
    A, Q, dm = make_problem(d=3, n=200, n_landmarks=6, n_ranges=650, seed=77)
    L = capi.load()
    rng = np.random.default_rng(1)
    X = rng.standard_normal((dm.N, p))
    Y = orc.project_manifold(dm, rng.uniform(-1, 1, (dm.N, p)))
    G = orc.egrad(Q, Y)
    V = orc.tangent_proj(dm, Y, rng.uniform(-1, 1, (dm.N, p)))
    for win in (0, 1 << 30):
        L.cora_debug_spmm_window_min_slices(win)
        c = ctx_for(Q, dm, p)
        assert relerr(c.dataMatrixProduct(X), orc.spmm(Q, X)) < REL
        assert relerr(c.Riemannian_Hessian_vector_product(Y, G, V), orc.hvp(Q, dm, Y, G, V)) < REL
        c.close()


def test_matrix_without_the_symmetries_keeps_the_plain_layout(window_form):
    """The chain layout relies on Q = Q^T in three places and checks each bit for bit when the format is built; a matrix
    that breaks one (the data-matrix product accepts any CSR) keeps the plain layout for the slices concerned and the
    product stays exact."""
    import scipy.sparse as sp
    A, Q, dm = make_problem(d=3, n=150, n_landmarks=2, n_ranges=100, seed=5)
    S = Q.to_scipy().tolil()
    tb = dm.dn + dm.r
    S[tb + 70, 3 * 70 + 1] *= 1.5          # Q31 != Q13^T on pose 70's own block  -> its slice falls back
    S[3 * 10 + 2, 3 * 9 + 1] += 0.25       # the block coupling poses 9 and 10 is no longer the transpose of its mirror
    S = S.tocsr()
    S.sort_indices()
    Q2 = orc.CSR.from_scipy(S)
    full = ctx_for(Q, dm, 5).format_stats()
    c = ctx_for(Q2, dm, 5)
    st = c.format_stats()
    assert st["slices"] > full["slices"]   # translation-row slices are back for the slices that fell back
    X = np.random.default_rng(2).standard_normal((dm.N, 5))
    assert relerr(c.dataMatrixProduct(X), orc.spmm(Q2, X)) < REL
    c.close()
