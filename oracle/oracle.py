"""ctypes wrapper around oracle/libcora_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module (see the header of cora_oracle.c).  All dense arrays are
column-major float64 (Fortran order), exactly like the reference's
Eigen::MatrixXd; Q is CSR with int32 indices.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "libcora_oracle.so")
    src = os.path.join(_HERE, "cora_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcora_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_cost.restype = C.c_double
        _LIB.orc_inner.restype = C.c_double
        _LIB.orc_chol_factor.restype = C.c_void_p
        _LIB.orc_chol_nnz.restype = C.c_longlong
    return _LIB


def set_threads(n):
    """Threads of the row-parallel oracle loops (1 = like the single-threaded reference)."""
    lib().orc_set_threads(int(n))


def max_threads():
    return int(lib().orc_max_threads())


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f(a):
    """column-major float64 2-D view/copy"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.asfortranarray(a)


class CSR:
    """Symmetric data matrix Q in CSR (int32 / float64), as Eigen stores it."""

    def __init__(self, rowptr, col, val, N):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.N = int(N)
        self.nnz = int(self.rowptr[-1]) if len(self.rowptr) else 0

    @staticmethod
    def from_scipy(A):
        A = A.tocsr()
        A.sum_duplicates()
        A.sort_indices()
        return CSR(A.indptr, A.indices, A.data, A.shape[0])

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col, self.rowptr), shape=(self.N, self.N))


class Dims:
    """(d, n poses, r range measurements, N) -- N = n(d+1) + l + r."""

    def __init__(self, d, n, r, N):
        self.d, self.n, self.r, self.N = int(d), int(n), int(r), int(N)
        self.dn = self.d * self.n
        self.n_trans = self.N - self.dn - self.r


def spmm(Q, X, rowwise=False):
    X = _f(X)
    out = np.empty((Q.N, X.shape[1]), order="F")
    fn = lib().orc_spmm_rowwise if rowwise else lib().orc_spmm
    fn(Q.N, _i(Q.rowptr), _i(Q.col), _d(Q.val), _d(X), X.shape[0], X.shape[1],
       _d(out), Q.N)
    return out


def cost(Q, Y):
    Y = _f(Y)
    work = np.empty((Q.N, Y.shape[1]), order="F")
    return lib().orc_cost(Q.N, _i(Q.rowptr), _i(Q.col), _d(Q.val), _d(Y),
                          Y.shape[0], Y.shape[1], _d(work))


def inner(A, B):
    A, B = _f(A), _f(B)
    return lib().orc_inner(A.shape[0], A.shape[1], _d(A), A.shape[0], _d(B), B.shape[0])


def egrad(Q, Y):
    return spmm(Q, Y)


def tangent_proj(dm, Y, V):
    Y, V = _f(Y), _f(V)
    out = np.empty_like(V, order="F")
    lib().orc_tangent_proj(dm.d, dm.n, dm.r, dm.N, Y.shape[1], _d(Y), dm.N,
                           _d(V), dm.N, _d(out), dm.N)
    return out


def rgrad(Q, dm, Y):
    return tangent_proj(dm, Y, egrad(Q, Y))


def _untouched(shape, dtype):
    """An array on freshly mapped pages (numpy's allocator may hand back memory another thread has touched)."""
    import mmap
    count = int(np.prod(shape))
    mm = mmap.mmap(-1, max(count * np.dtype(dtype).itemsize, 1))
    return np.frombuffer(mm, dtype=dtype, count=count).reshape(shape, order="F")


def cpu_order():
    """Logical CPUs ordered socket by socket, core by core, SMT siblings next to each other."""
    import glob, os
    cpus = []
    for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*"):
        try:
            c = int(os.path.basename(d)[3:])
            pk = int(open(d + "/topology/physical_package_id").read())
            co = int(open(d + "/topology/core_id").read())
        except (OSError, ValueError):
            continue
        cpus.append((pk, co, c))
    allowed = os.sched_getaffinity(0)
    return [c for _, _, c in sorted(cpus) if c in allowed]


def bind_threads(threads):
    """Spread the team over the machine: thread t on the (t * cpus / threads)-th CPU of cpu_order() -- neighbouring
    threads (neighbouring rows) on neighbouring cores, the first half of the rows on the first socket.  threads = 0
    lifts the binding.  Returns the CPUs used."""
    if threads <= 0:
        lib().orc_bind_threads(0, None)
        return []
    order = cpu_order()
    if not order:
        return []
    pick = np.array([order[(t * len(order)) // threads] if threads <= len(order) else order[t % len(order)]
                     for t in range(threads)], dtype=np.int32)
    lib().orc_bind_threads(len(pick), pick.ctypes.data_as(_ip))
    return pick.tolist()


def numa_csr(Q):
    """Copy of Q whose rows sit on the memory node of the thread that works on them (orc_first_touch_csr)."""
    out = CSR.__new__(CSR)
    out.rowptr, out.N, out.nnz = Q.rowptr, Q.N, Q.nnz
    out.col = _untouched((Q.nnz,), np.int32)
    out.val = _untouched((Q.nnz,), np.float64)
    lib().orc_first_touch_csr(Q.N, _i(Q.rowptr), _i(Q.col), _d(Q.val), _i(out.col), _d(out.val))
    return out


def numa_dense(A=None, shape=None):
    """Column-major copy of A (or zeros of `shape`) first touched with the threaded loops' row schedule."""
    if A is not None:
        A = _f(A)
        shape = A.shape
    out = _untouched(shape, np.float64)
    lib().orc_first_touch_dense(shape[0], shape[1], None if A is None else _d(A), shape[0], _d(out), shape[0])
    return out


def hvp(Q, dm, Y, G, Ydot, out=None, work=None):
    Y, G, Ydot = _f(Y), _f(G), _f(Ydot)
    p = Y.shape[1]
    if out is None:
        out = np.empty((dm.N, p), order="F")
    if work is None:
        work = np.empty((dm.N, p), order="F")
    lib().orc_hvp(dm.d, dm.n, dm.r, dm.N, p, _i(Q.rowptr), _i(Q.col), _d(Q.val),
                  _d(Y), dm.N, _d(G), dm.N, _d(Ydot), dm.N, _d(out), dm.N, _d(work))
    return out


def lambda_blocks(Q, dm, Y):
    Y = _f(Y)
    QY = spmm(Q, Y)
    Lst = np.zeros((dm.d, max(dm.dn, 1)), order="F")[:, :dm.dn]
    Lst = np.asfortranarray(Lst)
    lob = np.zeros(max(dm.r, 1))[:dm.r].copy()
    lib().orc_lambda_blocks(dm.d, dm.n, dm.r, Y.shape[1], _d(Y), dm.N, _d(QY),
                            dm.N, _d(Lst), _d(lob))
    return Lst, lob


def S_apply(Q, dm, Lst, lob, X):
    X = _f(X)
    k = X.shape[1]
    out = np.empty((dm.N, k), order="F")
    work = np.empty((dm.N, k), order="F")
    Lst = np.asfortranarray(Lst)
    lob = np.ascontiguousarray(lob)
    lib().orc_S_apply(dm.d, dm.n, dm.r, dm.N, _i(Q.rowptr), _i(Q.col), _d(Q.val),
                      _d(Lst), _d(lob), _d(X), dm.N, k, _d(out), dm.N, _d(work))
    return out


def certificate_matrix_dense(Q, dm, Y):
    """S = Q - Lambda as a dense matrix (small cases only).
    Problem::get_certificate_matrix, src/CORA_problem.cpp:1162-1166."""
    Lst, lob = lambda_blocks(Q, dm, Y)
    S = Q.to_scipy().toarray()
    d = dm.d
    for i in range(dm.n):
        S[i * d:(i + 1) * d, i * d:(i + 1) * d] -= Lst[:, i * d:(i + 1) * d]
    for j in range(dm.r):
        S[dm.dn + j, dm.dn + j] -= lob[j]
    return S


def project_manifold(dm, A):
    A = _f(A)
    out = np.empty_like(A, order="F")
    lib().orc_project_manifold(dm.d, dm.n, dm.r, dm.N, A.shape[1], _d(A), dm.N,
                               _d(out), dm.N)
    return out


def retract(dm, Y, V):
    Y, V = _f(Y), _f(V)
    out = np.empty_like(Y, order="F")
    lib().orc_retract(dm.d, dm.n, dm.r, dm.N, Y.shape[1], _d(Y), dm.N, _d(V),
                      dm.N, _d(out), dm.N)
    return out


def diag(Q):
    out = np.empty(Q.N)
    lib().orc_diag(Q.N, _i(Q.rowptr), _i(Q.col), _d(Q.val), _d(out))
    return out


def precond_jacobi(Q, dm, Y, V):
    Y, V = _f(Y), _f(V)
    dinv = 1.0 / diag(Q)
    out = np.empty_like(V, order="F")
    lib().orc_precond_jacobi(dm.d, dm.n, dm.r, dm.N, Y.shape[1], _d(dinv), _d(Y),
                             dm.N, _d(V), dm.N, _d(out), dm.N)
    return out


class Implicit:
    """Translation-implicit formulation, restated from src/CORA_problem.cpp:714-753 (matrices),
    :745-753 (product), :822-867 (Hessian-vector product with the implicit product),
    :1168-1197 (translation recovery).  Variables have dn + r rows."""

    def __init__(self, Q, dm):
        A = Q.to_scipy().tocsr()
        m = dm.dn + dm.r
        nt = dm.N - m
        self.dm_full = dm
        self.dm = Dims(dm.d, dm.n, dm.r, m)                 # manifold ops on the leading rows only
        self.Qmain = CSR.from_scipy(A[:m, :m].tocsr())       # :723-724
        self.B = A[:m, m:m + nt - 1].tocsr()                 # TransOffDiagRed_, :729-731
        self.chol = Cholesky(CSR.from_scipy(A[m:m + nt - 1, m:m + nt - 1].tocsr()))  # LtransCholRed_, :737-739
        assert self.chol.ok

    def product(self, Y):                                    # :747-752
        Y = _f(Y)
        P2 = self.chol.solve(np.asfortranarray(self.B.T @ Y))
        return np.asfortranarray(spmm(self.Qmain, Y) - self.B @ P2)

    def cost(self, Y):
        return 0.5 * float(np.sum(_f(Y) * self.product(Y)))

    def rgrad(self, Y):
        return tangent_proj(self.dm, Y, self.product(Y))

    def lambda_blocks(self, Y):
        Y = _f(Y)
        dm = self.dm
        QY = self.product(Y)
        Lst = np.asfortranarray(np.zeros((dm.d, max(dm.dn, 1)), order="F")[:, :dm.dn])
        lob = np.zeros(max(dm.r, 1))[:dm.r].copy()
        lib().orc_lambda_blocks(dm.d, dm.n, dm.r, Y.shape[1], _d(Y), dm.N, _d(QY), dm.N, _d(Lst), _d(lob))
        return Lst, lob

    def hvp(self, Y, Ydot):
        dm = self.dm
        Lst, lob = self.lambda_blocks(Y)
        H = self.product(Ydot)
        for i in range(dm.n):
            s = slice(i * dm.d, (i + 1) * dm.d)
            H[s] -= Lst[:, s] @ Ydot[s]
        H[dm.dn:dm.dn + dm.r] -= lob[:, None] * Ydot[dm.dn:dm.dn + dm.r]
        return tangent_proj(dm, Y, H)

    def translation_explicit(self, Y):                       # :1181-1192
        Y = _f(Y)
        t = -self.chol.solve(np.asfortranarray(self.B.T @ Y))
        return np.asfortranarray(np.vstack([Y, t, np.zeros((1, Y.shape[1]))]))


class Cholesky:
    """Sparse LL^T of a symmetric CSR matrix; `ok` False <=> not positive definite
    (the reference's `MChol.info() == Eigen::Success`, src/CORA_utils.cpp:51)."""

    def __init__(self, A, perm=None):
        self.n = A.N
        self._A = A
        p = None
        if perm is not None:
            self._perm = np.ascontiguousarray(perm, dtype=np.int32)
            p = _i(self._perm)
        self._h = lib().orc_chol_factor(A.N, _i(A.rowptr), _i(A.col), _d(A.val), p)
        self.ok = bool(self._h)

    @property
    def nnz(self):
        return lib().orc_chol_nnz(C.c_void_p(self._h)) if self.ok else -1

    def solve(self, B):
        B = _f(B)
        X = np.empty_like(B, order="F")
        lib().orc_chol_solve(C.c_void_p(self._h), _d(B), B.shape[0], B.shape[1],
                             _d(X), X.shape[0])
        return X

    def precond(self, dm, Y, V):
        Y, V = _f(Y), _f(V)
        out = np.zeros_like(V, order="F")
        lib().orc_precond_chol(C.c_void_p(self._h), dm.d, dm.n, dm.r, dm.N,
                               Y.shape[1], _d(Y), dm.N, _d(V), dm.N, _d(out), dm.N)
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_chol_free(C.c_void_p(self._h))
            self._h = None
