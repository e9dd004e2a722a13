"""Minimal MatrixMarket coordinate reader for the golden fixtures.

Mirrors the reference loader tests/test_utils.cpp:24-53: files whose banner
says `symmetric` hold the lower triangle and are mirrored."""
import numpy as np
import scipy.sparse as sp


def read_mm(path):
    with open(path) as fh:
        banner = fh.readline()
        sym = "symmetric" in banner
        line = fh.readline()
        while line.startswith("%"):
            line = fh.readline()
        rows, cols, nnz = (int(x) for x in line.split())
        i = np.empty(nnz, dtype=np.int64)
        j = np.empty(nnz, dtype=np.int64)
        v = np.empty(nnz)
        for k in range(nnz):
            a, b, c = fh.readline().split()
            i[k], j[k], v[k] = int(a) - 1, int(b) - 1, float(c)
    A = sp.coo_matrix((v, (i, j)), shape=(rows, cols)).tocsr()
    if sym:
        A = A + sp.tril(A, -1).T
    return sp.csr_matrix(A)


def read_dense(path):
    return np.asfortranarray(read_mm(path).toarray())
