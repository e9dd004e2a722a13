"""CPU decision of the certificate test for the GPU tests: S = Q - Lambda(X) assembled with scipy from the oracle's
Lambda blocks, the oracle's sparse Cholesky of S + eta I (the reference's criterion, src/CORA_utils.cpp:36-51)."""
import numpy as np
import scipy.sparse as sp

from oracle import oracle as orc


def elimination_order(Q, dims):
    """Fill-reducing order for the oracle's Cholesky: per pose its rotation rows, the range rows hanging off it and its
    translation; landmarks last (the idea of the host's coraOrdering, recomputed here so that the CPU check does not
    depend on the code under test).  Range rows between two poses hang off the first one."""
    A = Q.to_scipy().tocsr()
    d, n, r, N = dims.d, dims.n, dims.r, dims.N
    tb = d * n + r
    C = A[d * n:tb][:, tb:tb + n].tocsr()
    owner = np.full(r, -1, dtype=np.int64)
    has = np.diff(C.indptr) > 0
    owner[has] = C.indices[C.indptr[:-1][has]]
    key = np.empty(N, dtype=np.float64)
    for a in range(d):
        key[a:d * n:d] = np.arange(n) + 0.1 * a / d
    key[d * n:tb] = np.where(owner >= 0, owner + 0.5, n + 1.0)
    key[tb:tb + n] = np.arange(n) + 0.9
    key[tb + n:] = n + 2.0
    return np.argsort(key, kind="stable").astype(np.int32)


def certificate_matrix(Q, dims, X):
    """S = Q - Lambda(X) (src/CORA_problem.cpp:1105-1166) as a scipy CSR matrix."""
    Lst, lob = orc.lambda_blocks(Q, dims, X)
    d = dims.d
    blocks = sp.block_diag([Lst[:, d * i:d * i + d] for i in range(dims.n)], format="csr")
    Lam = sp.block_diag([blocks, sp.diags(lob), sp.csr_matrix((dims.N - dims.dn - dims.r,) * 2)], format="csr")
    return (Q.to_scipy() - Lam).tocsr()


def minimum_degree_order(S):
    """A fill-reducing elimination order for any pattern (several robots tied by ranges, loop closures): the minimum
    degree order SuperLU computes for A^T + A.  Only the ORDER is taken from there -- the factorisation that decides is
    the oracle's.  (On tiers / MR.CLAM the pose-major order above fills in every inter-robot range: 80 - 105 s per
    factorisation on one core against 0.3 s.)"""
    from scipy.sparse.linalg import splu
    n = S.shape[0]
    pat = (abs(S) + sp.identity(n) * (abs(S).max() * n + 1.0)).tocsc()   # same pattern, diagonally dominant: never singular
    lu = splu(pat, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options={"SymmetricMode": True})
    return np.argsort(lu.perm_c).astype(np.int32)


def oracle_is_certified(Q, dims, X, eta, perm=None):
    """True iff S(X) + eta I has a Cholesky factor."""
    S = (certificate_matrix(Q, dims, X) + eta * sp.identity(dims.N)).tocsr()
    S.sort_indices()
    return orc.Cholesky(orc.CSR.from_scipy(S), perm=minimum_degree_order(S) if perm is None else perm).ok
