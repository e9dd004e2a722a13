"""Device solve plan of a sparse Cholesky factor (cora_amd/csrc/trisolve.h) without a GPU:
`cora_debug_factor_solve_host` builds the stages and explicit block inverses and runs the products
on the host in launch order.  Checked against scipy on matrices with arbitrary structure -- not only
the pose chains the ordering is tuned for."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from cora_amd import capi

_ip = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)


def _factor_csc(A):
    """dense Cholesky -> CSC of L with the diagonal first (structural zeros stay exact zeros)."""
    L = np.linalg.cholesky(A.toarray())
    Ls = sp.csc_matrix(np.where(np.abs(L) > 0, L, 0.0))
    Ls.sort_indices()
    return Ls


def _solve(Ls, B):
    lib = capi.load()
    m = Ls.shape[0]
    Lp = np.ascontiguousarray(Ls.indptr, dtype=np.int32)
    Li = np.ascontiguousarray(Ls.indices, dtype=np.int32)
    Lx = np.ascontiguousarray(Ls.data, dtype=np.float64)
    B = np.asfortranarray(B, dtype=np.float64)
    X = np.zeros_like(B, order="F")
    st = (C.c_int64 * 4)()
    rc = lib.cora_debug_factor_solve_host(m, Lp.ctypes.data_as(_ip), Li.ctypes.data_as(_ip), Lx.ctypes.data_as(_dp),
                                          B.shape[1], B.ctypes.data_as(_dp), X.ctypes.data_as(_dp), st)
    assert rc == 0, lib.cora_last_error(None).decode()
    return X, dict(stages=st[0], nnzW=st[1], nnzL=st[2], dense_blocks=st[3])


def _spd(n, kind, rng):
    if kind == "chain":          # block tridiagonal, like an odometry chain
        A = sp.diags([np.full(n - k, -1.0 / (k + 1)) for k in range(1, 5)], list(range(1, 5)), shape=(n, n))
    elif kind == "arrow":        # chain + a few dense trailing rows (landmarks)
        A = sp.diags([np.full(n - 1, -1.0)], [1], shape=(n, n)).tolil()
        for r in range(n - 6, n):
            A[r, rng.choice(n - 6, size=(n - 6) // 2, replace=False)] = -0.01
        A = sp.triu(A.tocsr().T + A.tocsr(), 1)
    else:                        # random sparse graph Laplacian: no chain structure at all
        nz = 3 * n
        A = sp.coo_matrix((-rng.uniform(0.1, 1.0, nz), (rng.integers(0, n, nz), rng.integers(0, n, nz))), shape=(n, n))
        A = sp.triu(A.tocsr(), 1)
    A = (A + A.T).tocsr()
    d = np.asarray(abs(A).sum(axis=1)).ravel() + rng.uniform(0.1, 1.0, n)
    return (A + sp.diags(d)).tocsr()


@pytest.mark.parametrize("kind,n", [("chain", 40), ("chain", 3000), ("arrow", 2500), ("random", 700), ("random", 2600)])
def test_staged_plan_matches_direct_solve(kind, n):
    rng = np.random.default_rng(n)
    A = _spd(n, kind, rng)
    Ls = _factor_csc(A)
    B = rng.uniform(-1, 1, (n, 3))
    X, st = _solve(Ls, B)
    ref = np.linalg.solve(A.toarray(), B)
    assert np.abs(X - ref).max() < 1e-11 * np.abs(ref).max()
    assert st["nnzL"] == Ls.nnz and st["stages"] >= 1
    if kind == "chain" and n == 3000:     # a 3000-row path: its full inverse (4.5 M entries) is not worth forming
        assert st["stages"] >= 2
        assert st["dense_blocks"] > 0     # and its leaves fit the dense wavefront kernel
    if kind == "chain" and n == 40:
        assert st["stages"] == 1          # small factors are applied as one explicit inverse


def test_rejects_malformed_factor():
    lib = capi.load()
    # diagonal not first in column 0
    Lp = np.array([0, 2, 3], dtype=np.int32)
    Li = np.array([1, 0, 1], dtype=np.int32)
    Lx = np.array([0.5, 1.0, 1.0])
    B = np.ones((2, 1), order="F")
    X = np.zeros((2, 1), order="F")
    rc = lib.cora_debug_factor_solve_host(2, Lp.ctypes.data_as(_ip), Li.ctypes.data_as(_ip), Lx.ctypes.data_as(_dp), 1,
                                          B.ctypes.data_as(_dp), X.ctypes.data_as(_dp), None)
    assert rc != 0 and b"diagonal first" in lib.cora_last_error(None)


def test_incomplete_factor_goes_through_the_staged_plan():
    """An INCOMPLETE factor (entries dropped, as the ILDL preconditioner of fast_verification produces) no longer has
    the elimination tree of a complete one; the plan takes the tree of the factor's own pattern.  The staged
    products must apply (L L^T)^-1 of exactly the factor that was given."""
    import scipy.sparse.linalg as spl
    rng = np.random.default_rng(8)
    n = 2600
    A = _spd(n, "random", rng)
    Ls = _factor_csc(A).tolil()
    L = Ls.tocoo()
    keep = (L.row == L.col) | (np.abs(L.data) > 0.02) | (rng.uniform(size=L.nnz) < 0.3)
    Li = sp.csc_matrix((L.data[keep], (L.row[keep], L.col[keep])), shape=(n, n))
    Li.sort_indices()
    assert Li.nnz < 0.8 * L.nnz
    B = rng.uniform(-1, 1, (n, 2))
    X, st = _solve(Li, B)
    Y = spl.spsolve_triangular(Li.tocsr(), B, lower=True)
    ref = spl.spsolve_triangular(Li.T.tocsr(), Y, lower=False)
    assert np.abs(X - ref).max() < 1e-10 * np.abs(ref).max()
    assert st["nnzL"] == Li.nnz


@pytest.mark.parametrize("kind,n", [("chain", 3000), ("chain", 9000), ("arrow", 2500)])
def test_two_stage_plan_with_the_aux_sums_as_their_own_product(kind, n, monkeypatch):
    """Two-stage plans add the substitution blocks' couplings (aux rows) to the top stage's right-hand side either
    inside the top product (entries repeated per aux row) or -- large top stages -- as a product of their own in the
    slot of the unused "a" product.  Forced here (CORA_TRI_UNFOLD_MIN=0); both forms must solve the system."""
    rng = np.random.default_rng(n + 1)
    A = _spd(n, kind, rng)
    Ls = _factor_csc(A)
    B = rng.uniform(-1, 1, (n, 2))
    ref = np.linalg.solve(A.toarray(), B)
    monkeypatch.setenv("CORA_TRI_UNFOLD_MIN", "0")
    X, st = _solve(Ls, B)
    assert np.abs(X - ref).max() < 1e-11 * np.abs(ref).max()
    monkeypatch.setenv("CORA_TRI_UNFOLD_MIN", "1000000000")
    X2, st2 = _solve(Ls, B)
    assert np.abs(X2 - ref).max() < 1e-11 * np.abs(ref).max()
    assert st["stages"] == st2["stages"]
