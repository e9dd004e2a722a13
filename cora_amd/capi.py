"""ctypes binding of libcora_hip.so (include/cora_hip.h).

Plumbing for tests/ and bench.py only: the product is the shared library and
the C++ host behind it.  There is no fallback -- if the library is missing or
no gfx950 device is usable, calls raise CoraError."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_LIB = None

STATUS = {0: "CORA_OK", 1: "CORA_ERR_SHAPE", 2: "CORA_ERR_NOT_READY", 3: "CORA_ERR_NAN",
          4: "CORA_ERR_HIP", 5: "CORA_ERR_ARG", 6: "CORA_ERR_NOMEM"}
PRECOND_NONE, PRECOND_JACOBI, PRECOND_BLOCK_CHOLESKY, PRECOND_REGULARIZED_CHOLESKY = 0, 1, 2, 3


# callback types of cora_set_comm (user pointer, device / host pointer as an integer, count)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


class CoraError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS.get(code, str(code)), msg))
        self.code = code


def lib_path():
    return _build.LIB


def load():
    """Loads the in-tree shared library (building it first when stale)."""
    global _LIB
    if _LIB is None:
        # torch bundles its own libamdhip64.so.7: load it FIRST so that the process has one
        # HIP runtime (our library then binds to the already-loaded SONAME); the other order
        # leaves two runtimes and hipSetDevice fails with "no ROCm-capable device".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = _build.LIB
        # lab builds of one kernel group (cora_amd/build.py, CORA_VARIANT): never set outside measurement scripts
        variant = os.environ.get("CORA_LIB_VARIANT")
        if variant:
            path = os.path.join(_build.LIBDIR, "variants", variant, "libcora_hip.so")
            if not os.path.exists(path):
                raise RuntimeError("CORA_LIB_VARIANT=%s: %s does not exist" % (variant, path))
        elif _build.needs_build():
            try:
                _build.build()
            except Exception as e:  # hipcc missing on a box that ships the prebuilt .so
                if not os.path.exists(path):
                    raise RuntimeError("libcora_hip.so is missing and cannot be built: %s" % e)
        _LIB = C.CDLL(path)
        L = _LIB
        L.cora_last_error.restype = C.c_char_p
        L.cora_last_error.argtypes = [C.c_void_p]
        for name in ("cora_rows", "cora_shard_rows", "cora_shard_begin", "cora_nnz", "cora_dim"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("cora_point_Y_dev", "cora_point_egrad_dev", "cora_point_rgrad_dev"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.cora_ctx_destroy.restype = None
        L.cora_ctx_destroy.argtypes = [C.c_void_p]
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.asfortranarray(a)


class Context:
    """One device-resident problem (cora_ctx).  Host arrays are column-major
    N x k float64, as in the reference (Eigen::MatrixXd)."""

    def __init__(self, d, n_poses, n_ranges, n_trans, rowptr, colidx, vals, device=0, rank=0, world=1,
                 whole_long_rows=False):
        self.L = load()
        self.d, self.n, self.r, self.nt = int(d), int(n_poses), int(n_ranges), int(n_trans)
        self.N = self.d * self.n + self.r + self.nt
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        if len(rowptr) != self.N + 1:
            raise CoraError(1, "rowptr must have N+1 entries")
        h = C.c_void_p()
        rc = self.L.cora_ctx_create_part_opts(C.c_int(device), self.d, self.n, self.r, self.nt,
                                              rowptr.ctypes.data_as(_ip), colidx.ctypes.data_as(_ip), _d(vals),
                                              C.c_int(rank), C.c_int(world), C.c_uint(1 if whole_long_rows else 0),
                                              C.byref(h))
        if rc:
            raise CoraError(rc, self.L.cora_last_error(None).decode())
        self.h = h
        self.p = 0

    @classmethod
    def from_handle(cls, ptr, d, n_poses, n_ranges, n_trans):
        """Borrow a cora_ctx owned by someone else (e.g. CORA::Problem::context()); never destroyed here."""
        self = cls.__new__(cls)
        self.L = load()
        self.d, self.n, self.r, self.nt = int(d), int(n_poses), int(n_ranges), int(n_trans)
        self.N = self.d * self.n + self.r + self.nt
        self.h = C.c_void_p(ptr)
        self.p = self.L.cora_get_rank(self.h)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.L.cora_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise CoraError(rc, self.L.cora_last_error(self.h).decode())

    # ---- queries
    def set_rank(self, p):
        self._chk(self.L.cora_set_rank(self.h, int(p)))
        self.p = int(p)

    @property
    def ld(self):
        return self.L.cora_ld(self.h)

    @property
    def rows(self):
        return self.L.cora_rows(self.h)

    @property
    def shard_rows(self):
        return self.L.cora_shard_rows(self.h)

    @property
    def shard_begin(self):
        return self.L.cora_shard_begin(self.h)

    @property
    def nnz(self):
        return self.L.cora_nnz(self.h)

    def row_map(self):
        m = np.empty(self.N, dtype=np.int32)
        self._chk(self.L.cora_row_map(self.h, m.ctypes.data_as(_ip)))
        return m

    def remote_rows(self):
        """Internal rows outside this rank's shard that its part of Q reads (ascending, int64)."""
        n = C.c_int64()
        self._chk(self.L.cora_remote_rows(self.h, None, C.byref(n)))
        r = np.empty(max(n.value, 1), dtype=np.int32)
        self._chk(self.L.cora_remote_rows(self.h, r.ctypes.data_as(_ip), C.byref(n)))
        return r[:n.value].astype(np.int64)

    def long_rows(self):
        """API rows of the distributed long rows of a partitioned handle (empty on one GPU)."""
        n = C.c_int64(0)
        self._chk(self.L.cora_long_rows(self.h, None, C.byref(n)))
        r = np.zeros(max(n.value, 1), dtype=np.int32)
        self._chk(self.L.cora_long_rows(self.h, r.ctypes.data_as(_ip), C.byref(n)))
        return r[:n.value].astype(np.int64)

    def format_stats(self):
        s = (C.c_int64 * 8)()
        self._chk(self.L.cora_format_stats(self.h, s))
        keys = ["slices", "padded_nnz", "long_nnz", "long_rows", "long_chunks", "local_rows", "local_nnz",
                "max_width"]
        return dict(zip(keys, [int(x) for x in s]))

    def format_bytes(self):
        b = (C.c_int64 * 4)()
        self._chk(self.L.cora_format_bytes(self.h, b))
        return dict(zip(["values", "indices", "descriptors", "total"], [int(x) for x in b]))

    def precond_entries(self):
        """Entries the installed preconditioner's solve plan stores (cora_precond_entries)."""
        s = (C.c_int64 * 6)()
        self._chk(self.L.cora_precond_entries(self.h, s))
        keys = ["top_forward", "top_backward", "sub_forward_slots", "sub_backward_slots", "sub_blocks", "aux_rows"]
        return dict(zip(keys, [int(x) for x in s]))

    def set_stream(self, stream_ptr):
        self._chk(self.L.cora_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- multi-GPU building blocks (include/cora_hip.h, cora_set_comm)
    @property
    def rank(self):
        return self.L.cora_rank(self.h)

    @property
    def world(self):
        return self.L.cora_world(self.h)

    def set_comm(self, exchange, allreduce, allgather):
        """ctypes callbacks (EXCHANGE_FN / ALLREDUCE_FN / ALLGATHER_FN); the caller keeps them alive."""
        self._chk(self.L.cora_set_comm(self.h, exchange, allreduce, allgather, None))

    def pack_rows_dev(self, x, ld, rows_ptr, n, packed):
        self._chk(self.L.cora_pack_rows_dev(self.h, C.c_void_p(x), int(ld), C.c_void_p(rows_ptr), C.c_int64(n),
                                            C.c_void_p(packed)))

    def scatter_rows_dev(self, packed, ld, rows_ptr, n, x):
        self._chk(self.L.cora_scatter_rows_dev(self.h, C.c_void_p(packed), int(ld), C.c_void_p(rows_ptr), C.c_int64(n),
                                               C.c_void_p(x)))

    def copy_rows_dev(self, src, ld, rows_ptr, n, dst):
        self._chk(self.L.cora_copy_rows_dev(self.h, C.c_void_p(src), int(ld), C.c_void_p(rows_ptr), C.c_int64(n),
                                            C.c_void_p(dst)))

    def copy_shard_dev(self, src, ld, shard, dst):
        self._chk(self.L.cora_copy_shard_dev(self.h, C.c_void_p(src), int(ld), int(shard), C.c_void_p(dst)))

    # ---- host-pointer operator API (mirrors CORA::Problem)
    def _out(self, k):
        return np.zeros((self.N, k), order="F")

    def dataMatrixProduct(self, X):
        X = _f(X)
        out = self._out(X.shape[1])
        self._chk(self.L.cora_data_matrix_product(self.h, _d(X), X.shape[0], X.shape[1], _d(out), self.N))
        return out

    def evaluateObjective(self, Y):
        Y = _f(Y)
        f = C.c_double()
        self._chk(self.L.cora_evaluate_objective(self.h, _d(Y), Y.shape[0], C.byref(f)))
        return f.value

    def Euclidean_gradient(self, Y):
        Y = _f(Y)
        out = self._out(self.p)
        self._chk(self.L.cora_euclidean_gradient(self.h, _d(Y), Y.shape[0], _d(out), self.N))
        return out

    def Riemannian_gradient(self, Y):
        Y = _f(Y)
        out = self._out(self.p)
        self._chk(self.L.cora_riemannian_gradient(self.h, _d(Y), Y.shape[0], _d(out), self.N))
        return out

    def tangent_space_projection(self, Y, V):
        Y, V = _f(Y), _f(V)
        out = self._out(self.p)
        self._chk(self.L.cora_tangent_space_projection(self.h, _d(Y), Y.shape[0], _d(V), V.shape[0], _d(out),
                                                       self.N))
        return out

    def Riemannian_Hessian_vector_product(self, Y, G, Ydot):
        Y, G, Ydot = _f(Y), _f(G), _f(Ydot)
        out = self._out(self.p)
        self._chk(self.L.cora_riemannian_hessian_vector_product(
            self.h, _d(Y), Y.shape[0], _d(G), G.shape[0], _d(Ydot), Ydot.shape[0], _d(out), self.N))
        return out

    def projectToManifold(self, A):
        A = _f(A)
        out = self._out(self.p)
        self._chk(self.L.cora_project_to_manifold(self.h, _d(A), A.shape[0], _d(out), self.N))
        return out

    def retract(self, Y, V):
        Y, V = _f(Y), _f(V)
        out = self._out(self.p)
        self._chk(self.L.cora_retract(self.h, _d(Y), Y.shape[0], _d(V), V.shape[0], _d(out), self.N))
        return out

    def precond_setup(self, kind):
        self._chk(self.L.cora_precond_setup(self.h, int(kind)))

    def precondition(self, V):
        V = _f(V)
        out = self._out(self.p)
        self._chk(self.L.cora_precondition(self.h, _d(V), V.shape[0], _d(out), self.N))
        return out

    def compute_Lambda_blocks(self, Y):
        Y = _f(Y)
        st = np.zeros((self.d, max(self.d * self.n, 1)), order="F")
        ob = np.zeros(max(self.r, 1))
        self._chk(self.L.cora_compute_lambda_blocks(self.h, _d(Y), Y.shape[0], _d(st), _d(ob)))
        return st[:, :self.d * self.n], ob[:self.r]

    def set_point(self, Y):
        Y = _f(Y)
        self._chk(self.L.cora_set_point(self.h, _d(Y), Y.shape[0]))

    def point_cost(self):
        f = C.c_double()
        self._chk(self.L.cora_point_cost(self.h, C.byref(f)))
        return f.value

    def certificate_product(self, X):
        X = _f(X)
        out = self._out(X.shape[1])
        self._chk(self.L.cora_certificate_product(self.h, _d(X), X.shape[0], X.shape[1], _d(out), self.N))
        return out

    def inner_product(self, A, B):
        A, B = _f(A), _f(B)
        v = C.c_double()
        self._chk(self.L.cora_inner_product(self.h, _d(A), A.shape[0], _d(B), B.shape[0], A.shape[1],
                                            C.byref(v)))
        return v.value

    # ---- resident API (raw device pointers as ints)
    def dev_alloc(self, k):
        p = _dp()
        self._chk(self.L.cora_dev_alloc(self.h, int(k), C.byref(p)))
        return C.cast(p, C.c_void_p).value

    def dev_free(self, ptr):
        self._chk(self.L.cora_dev_free(self.h, C.c_void_p(ptr)))

    def upload(self, host, ptr):
        host = _f(host)
        self._chk(self.L.cora_upload(self.h, _d(host), host.shape[0], host.shape[1], C.c_void_p(ptr)))

    def download(self, ptr, k):
        out = self._out(k)
        self._chk(self.L.cora_download(self.h, C.c_void_p(ptr), int(k), _d(out), self.N))
        return out

    def set_point_dev(self, ptr):
        self._chk(self.L.cora_set_point_dev(self.h, C.c_void_p(ptr)))

    def spmm_dev(self, x, k, out):
        self._chk(self.L.cora_spmm_dev(self.h, C.c_void_p(x), int(k), C.c_void_p(out)))

    def hvp_dev(self, x, out):
        self._chk(self.L.cora_hvp_dev(self.h, C.c_void_p(x), C.c_void_p(out)))

    def certificate_product_dev(self, x, k, out):
        self._chk(self.L.cora_certificate_product_dev(self.h, C.c_void_p(x), int(k), C.c_void_p(out)))

    def tangent_space_projection_dev(self, v, out):
        self._chk(self.L.cora_tangent_space_projection_dev(self.h, C.c_void_p(v), C.c_void_p(out)))

    def precondition_projected_dev(self, v, out):
        self._chk(self.L.cora_precondition_projected_dev(self.h, C.c_void_p(v), C.c_void_p(out)))

    def retract_dev(self, v, alpha, out):
        self._chk(self.L.cora_retract_dev(self.h, C.c_void_p(v), C.c_double(alpha), C.c_void_p(out)))

    def objective_dev(self, y):
        """f(y) for a resident y without changing the current point."""
        f = C.c_double()
        self._chk(self.L.cora_objective_dev(self.h, C.c_void_p(y), C.byref(f)))
        return f.value

    def tnt_trial_dev(self, s, hs, xprop):
        """hs = Hess(s), xprop = Retr_Y(s); returns [<grad, s>, <s, Hess s>, <s, s>, f(xprop)] in one wait."""
        out = (C.c_double * 4)()
        self._chk(self.L.cora_tnt_trial_dev(self.h, C.c_void_p(s), C.c_void_p(hs), C.c_void_p(xprop), out))
        return [out[i] for i in range(4)]

    def tnt_accept_dev(self, x, pg):
        """x becomes the current point, pg = projected preconditioned gradient; returns [f, <g,g>, <Pg,Pg>, <g,Pg>]."""
        out = (C.c_double * 4)()
        self._chk(self.L.cora_tnt_accept_dev(self.h, C.c_void_p(x), C.c_void_p(pg), out))
        return [out[i] for i in range(4)]

    def project_to_manifold_dev(self, a, out):
        self._chk(self.L.cora_project_to_manifold_dev(self.h, C.c_void_p(a), C.c_void_p(out)))

    def axpby_dev(self, a, x, b, y):
        self._chk(self.L.cora_axpby_dev(self.h, C.c_double(a), C.c_void_p(x), C.c_double(b), C.c_void_p(y)))

    def axpy2_dev(self, a1, x1, y1, a2, x2, y2):
        self._chk(self.L.cora_axpy2_dev(self.h, C.c_double(a1), C.c_void_p(x1), C.c_void_p(y1), C.c_double(a2),
                                        C.c_void_p(x2), C.c_void_p(y2)))

    def stpcg_dev(self, grad, Delta, s, r, v, p, hp, kappa_fgr=0.1, theta=0.8, max_iters=80):
        """Device-resident Steihaug-Toint PCG at the current point; returns (Hessian-vector products, ||s||_M)."""
        it = C.c_int()
        sm = C.c_double()
        self._chk(self.L.cora_stpcg_dev(self.h, C.c_void_p(grad), C.c_double(Delta), C.c_double(kappa_fgr),
                                        C.c_double(theta), int(max_iters), C.c_void_p(s), C.c_void_p(r), C.c_void_p(v),
                                        C.c_void_p(p), C.c_void_p(hp), C.byref(it), C.byref(sm)))
        return it.value, sm.value

    def stpcg_warm_dev(self, grad, pg, g_g, g_pg, Delta, s, r, v, p, hp, kappa_fgr=0.1, theta=0.8, max_iters=80):
        """stpcg_dev started from a known preconditioned gradient pg = P grad, <grad, grad> and <grad, pg>."""
        it = C.c_int()
        sm = C.c_double()
        self._chk(self.L.cora_stpcg_warm_dev(self.h, C.c_void_p(grad), C.c_void_p(pg), C.c_double(g_g), C.c_double(g_pg),
                                             C.c_double(Delta), C.c_double(kappa_fgr), C.c_double(theta), int(max_iters),
                                             C.c_void_p(s), C.c_void_p(r), C.c_void_p(v), C.c_void_p(p), C.c_void_p(hp),
                                             C.byref(it), C.byref(sm)))
        return it.value, sm.value

    def profile_stpcg(self, on=True):
        self._chk(self.L.cora_debug_profile_stpcg(self.h, int(on)))

    def stpcg_path(self):
        """Iteration form of the last stpcg_dev call: 0 unfused, 1 fused vector passes, 2 sweep-fused."""
        return int(self.L.cora_debug_stpcg_path(self.h))

    def stpcg_graph_stats(self):
        """(graphs captured, batches replayed) of the device-resident STPCG (cora_debug_stpcg_graph)."""
        out = (C.c_long * 2)()
        self._chk(self.L.cora_debug_stpcg_graph(self.h, out))
        return int(out[0]), int(out[1])

    def stpcg_hvp_us(self):
        """(mean microseconds, count) of the Hessian-vector products of the last stpcg_dev call (profile_stpcg on)."""
        us, cnt = C.c_double(), C.c_int()
        self._chk(self.L.cora_debug_stpcg_hvp_us(self.h, C.byref(us), C.byref(cnt)))
        return us.value, cnt.value

    def stpcg_phase_us(self):
        """Mean microseconds per launch of the sweep-fused iteration of the last stpcg_dev call (profile_stpcg(2)):
        dict product | kappa | forward_sweep | top_forward | top_backward | backward_sweep; None where not recorded."""
        us = (C.c_double * 8)()
        self._chk(self.L.cora_debug_stpcg_phase_us(self.h, us))
        names = ("product", "kappa", "forward_sweep", "top_forward", "top_backward", "backward_sweep", "event_overhead")
        out = {k: (float(v) if v >= 0 else None) for k, v in zip(names, us)}
        out["kappa_folded"] = bool(us[7])
        if out["kappa_folded"]:
            out["kappa"] = None
        return out

    def gram_dev(self, a, ka, b, kb):
        """G = A^T B (ka x kb) of two resident blocks."""
        G = np.zeros((ka, kb), order="F")
        self._chk(self.L.cora_gram_dev(self.h, C.c_void_p(a), int(ka), C.c_void_p(b), int(kb),
                                       G.ctypes.data_as(C.POINTER(C.c_double))))
        return G

    def gram_batch_dev(self, pairs):
        """[A^T B for (a, ka, b, kb) in pairs] with one synchronisation (cora_gram_batch_dev)."""
        n = len(pairs)
        Gs = [np.zeros((ka, kb), order="F") for _, ka, _, kb in pairs]
        pa = (C.c_void_p * n)(*[C.c_void_p(a) for a, _, _, _ in pairs])
        pb = (C.c_void_p * n)(*[C.c_void_p(b) for _, _, b, _ in pairs])
        ka = (C.c_int * n)(*[int(k) for _, k, _, _ in pairs])
        kb = (C.c_int * n)(*[int(k) for _, _, _, k in pairs])
        out = (C.c_void_p * n)(*[G.ctypes.data_as(C.c_void_p) for G in Gs])
        self._chk(self.L.cora_gram_batch_dev(self.h, n, pa, ka, pb, kb, out))
        return Gs

    def combine_dev(self, xs, ks, Cs, kout, out):
        """out = sum_i X_i C_i with host coefficient matrices C_i (k_i x kout)."""
        n = len(xs)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(x) for x in xs])
        kk = (C.c_int * n)(*[int(k) for k in ks])
        mats = [np.asfortranarray(np.array(M, dtype=np.float64)) for M in Cs]
        cp = (C.c_void_p * n)(*[M.ctypes.data_as(C.c_void_p) for M in mats])
        self._chk(self.L.cora_combine_dev(self.h, n, ptrs, kk, cp, int(kout), C.c_void_p(out)))

    def dot_dev(self, a, b, k):
        v = C.c_double()
        self._chk(self.L.cora_dot_dev(self.h, C.c_void_p(a), C.c_void_p(b), int(k), C.byref(v)))
        return v.value

    def dots_dev(self, pairs):
        """Several inner products with one reduction and one synchronisation (the three metric() calls of
        an STPCG iteration, src/CORA.cpp:119-122)."""
        n = len(pairs)
        A = (C.c_void_p * n)(*[C.c_void_p(a) for a, _ in pairs])
        B = (C.c_void_p * n)(*[C.c_void_p(b) for _, b in pairs])
        out = (C.c_double * n)()
        self._chk(self.L.cora_dots_dev(self.h, n, A, B, out))
        return [out[i] for i in range(n)]

    def point_ptrs(self):
        return (self.L.cora_point_Y_dev(self.h), self.L.cora_point_egrad_dev(self.h),
                self.L.cora_point_rgrad_dev(self.h))

    def timer_start(self):
        self._chk(self.L.cora_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._chk(self.L.cora_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def sync(self):
        self._chk(self.L.cora_sync(self.h))

    # ---- test hook
    def debug_format_spmm_host(self, X):
        X = _f(X)
        out = self._out(X.shape[1])
        self._chk(self.L.cora_debug_format_spmm_host(self.h, _d(X), X.shape[0], X.shape[1], _d(out), self.N))
        return out
