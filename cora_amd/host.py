"""ctypes binding of the C++ host (include/cora_host.h) -- plumbing for tests/ and bench.py."""
import ctypes as C

import numpy as np

from . import capi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class HostError(RuntimeError):
    pass


def _lib():
    L = capi.load()
    if not getattr(L, "_host_ready", False):
        L.cora_host_last_error.restype = C.c_char_p
        L.cora_problem_destroy.restype = None
        L.cora_problem_destroy.argtypes = [C.c_void_p]
        L.cora_problem_context.restype = C.c_void_p
        L.cora_problem_context.argtypes = [C.c_void_p]
        L._host_ready = True
    return L


class Problem:
    """CORA::Problem of the C++ host."""

    def __init__(self, handle):
        self.L = _lib()
        self.h = handle

    @staticmethod
    def from_pyfg(path):
        L = _lib()
        h = C.c_void_p()
        if L.cora_problem_from_pyfg(path.encode(), C.byref(h)):
            raise HostError(L.cora_host_last_error().decode())
        return Problem(h)

    @staticmethod
    def synthetic(dim=3, n_poses=1000, n_landmarks=10, n_ranges=500, n_loops=0, seed=42,
                  precond=capi.PRECOND_JACOBI, pyfg_out=None, sigmas=None, ground_truth=False):
        """SURVEY 8(d) generator.  sigmas = (sigma_t, sigma_R, sigma_range) overrides the noise;
        ground_truth=True returns (problem, X_gt) with X_gt the N x dim truth in the explicit layout."""
        L = _lib()
        h = C.c_void_p()
        sg = np.ascontiguousarray(sigmas, dtype=np.float64) if sigmas is not None else None
        gt = np.zeros(((dim + 1) * n_poses + n_landmarks + n_ranges, dim), order="F") if ground_truth else None
        rc = L.cora_problem_synthetic_ex(dim, n_poses, n_landmarks, n_ranges, n_loops, C.c_uint64(seed), precond,
                                         sg.ctypes.data_as(_dp) if sg is not None else None,
                                         pyfg_out.encode() if pyfg_out else None,
                                         gt.ctypes.data_as(_dp) if gt is not None else None, C.byref(h))
        if rc:
            raise HostError(L.cora_host_last_error().decode())
        return (Problem(h), gt) if ground_truth else Problem(h)

    @staticmethod
    def new(dim, rank=None, implicit=False, precond=capi.PRECOND_REGULARIZED_CHOLESKY):
        """Empty CORA::Problem to be filled with add_* (the reference's programmatic API)."""
        L = _lib()
        h = C.c_void_p()
        if L.cora_problem_new(int(dim), int(rank or dim), int(bool(implicit)), int(precond), C.byref(h)):
            raise HostError(L.cora_host_last_error().decode())
        return Problem(h)

    @staticmethod
    def _m(a):
        a = np.asfortranarray(np.asarray(a, dtype=np.float64))
        return a, a.ctypes.data_as(_dp)

    def add_pose(self, sym):
        self._chk(self.L.cora_problem_add_pose(self.h, sym.encode()))

    def add_landmark(self, sym):
        self._chk(self.L.cora_problem_add_landmark(self.h, sym.encode()))

    def add_range(self, a, b, dist, cov):
        self._chk(self.L.cora_problem_add_range(self.h, a.encode(), b.encode(), C.c_double(dist), C.c_double(cov)))

    def add_rel_pose(self, a, b, R, t, cov):
        (_, pr), (_, pt), (_, pc) = self._m(R), self._m(t), self._m(cov)
        self._chk(self.L.cora_problem_add_rel_pose(self.h, a.encode(), b.encode(), pr, pt, pc))

    def add_rel_pose_landmark(self, a, b, t, cov):
        (_, pt), (_, pc) = self._m(t), self._m(cov)
        self._chk(self.L.cora_problem_add_rel_pose_landmark(self.h, a.encode(), b.encode(), pt, pc))

    def add_pose_prior(self, sym, R, t, cov):
        (_, pr), (_, pt), (_, pc) = self._m(R), self._m(t), self._m(cov)
        self._chk(self.L.cora_problem_add_pose_prior(self.h, sym.encode(), pr, pt, pc))

    def add_landmark_prior(self, sym, pos, cov):
        (_, pp), (_, pc) = self._m(pos), self._m(cov)
        self._chk(self.L.cora_problem_add_landmark_prior(self.h, sym.encode(), pp, pc))

    def close(self):
        if getattr(self, "h", None):
            self.L.cora_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise HostError(self.L.cora_host_last_error().decode())

    def update(self):
        self._chk(self.L.cora_problem_update(self.h))

    def dims(self):
        d = (C.c_int64 * 8)()
        self._chk(self.L.cora_problem_dims(self.h, d))
        keys = ["d", "n", "l", "r", "N", "nnz", "rpm", "rank"]
        return dict(zip(keys, [int(x) for x in d]))

    def matrix(self, name="DataMatrix"):
        """(rows, cols, rowptr, colidx, vals) as numpy copies."""
        rows, cols, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        rp, ci, va = _ip(), _ip(), _dp()
        self._chk(self.L.cora_problem_matrix(self.h, name.encode(), C.byref(rows), C.byref(cols), C.byref(nnz),
                                             C.byref(rp), C.byref(ci), C.byref(va)))
        n = nnz.value
        rowptr = np.ctypeslib.as_array(rp, shape=(rows.value + 1,)).copy()
        colidx = np.ctypeslib.as_array(ci, shape=(max(n, 1),))[:n].copy() if n else np.zeros(0, np.int32)
        vals = np.ctypeslib.as_array(va, shape=(max(n, 1),))[:n].copy() if n else np.zeros(0)
        return rows.value, cols.value, rowptr, colidx, vals

    def certificate_matrix(self, Y):
        """Problem::get_certificate_matrix: S = Q - Lambda(Y) as a scipy CSR matrix."""
        import scipy.sparse as sp
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
        rows, nnz = C.c_int64(), C.c_int64()
        rp, ci, va = _ip(), _ip(), _dp()
        self._chk(self.L.cora_problem_certificate_matrix(self.h, Y.ctypes.data_as(_dp), Y.shape[0], C.byref(rows),
                                                         C.byref(nnz), C.byref(rp), C.byref(ci), C.byref(va)))
        n = nnz.value
        return sp.csr_matrix((np.ctypeslib.as_array(va, shape=(n,)).copy(), np.ctypeslib.as_array(ci, shape=(n,)).copy(),
                              np.ctypeslib.as_array(rp, shape=(rows.value + 1,)).copy()), shape=(rows.value, rows.value))

    def scipy_matrix(self, name="DataMatrix"):
        import scipy.sparse as sp
        r, c, rp, ci, va = self.matrix(name)
        return sp.csr_matrix((va, ci, rp), shape=(r, c))

    def set_rank(self, p):
        self._chk(self.L.cora_problem_set_rank(self.h, int(p)))

    def set_preconditioner(self, kind):
        self._chk(self.L.cora_problem_set_preconditioner(self.h, int(kind)))

    def set_formulation(self, implicit):
        """Formulation::Implicit (translations marginalised) when true, Explicit otherwise."""
        self._chk(self.L.cora_problem_set_formulation(self.h, int(bool(implicit))))

    def variable_size(self):
        """Problem::getExpectedVariableSize(): N when explicit, d*n + r when implicit."""
        rows = C.c_int64()
        self._chk(self.L.cora_problem_variable_size(self.h, C.byref(rows)))
        return rows.value

    def set_device(self, dev):
        self._chk(self.L.cora_problem_set_device(self.h, int(dev)))

    def set_partition(self, rank, world, make_comm):
        """Problem::setPartition: this process (or thread) owns partition `rank` of `world` of the rows of Q.
        `make_comm(ctx)` builds the communicator (cora_amd.dist.TorchComm / ThreadComm) once the partitioned handle
        exists; it installs its callbacks on the handle itself.  Collective: every rank calls it."""
        from . import capi
        none = (capi.EXCHANGE_FN(), capi.ALLREDUCE_FN(), capi.ALLGATHER_FN())
        self._chk(self.L.cora_problem_set_partition(self.h, int(rank), int(world), none[0], none[1], none[2], None))
        dm = self.dims()
        ctx = capi.Context.from_handle(self.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
        self._comm = make_comm(ctx)
        return self._comm

    def op(self, name, A=None, B=None, C_=None):
        dm = self.dims()
        N, p = self.variable_size(), dm["rank"]

        def ptr(x):
            if x is None:
                return None
            x = np.asfortranarray(np.asarray(x, dtype=np.float64))
            if x.shape[0] != N:
                raise HostError("expected %d rows, got %d" % (N, x.shape[0]))
            cols.append(x.shape[1])
            keep.append(x)
            return x.ctypes.data_as(_dp)

        keep, cols = [], []
        if name == "evaluateObjective":
            out = np.zeros(1)
        pa, pb, pc = ptr(A), ptr(B), ptr(C_)
        if name != "evaluateObjective":
            full = name in ("getOdomInitialization", "getTranslationExplicitSolution", "alignEstimateToOrigin")
            out = np.zeros((dm["N"] if full else N, cols[0] if cols else p), order="F")
        if len(set(cols)) > 1:
            raise HostError("operands have different column counts: %s" % cols)
        self._chk(self.L.cora_problem_op(self.h, name.encode(), cols[0] if cols else p, pa, pb, pc,
                                         out.ctypes.data_as(_dp)))
        return float(out[0]) if name == "evaluateObjective" else out

    def lambda_blocks(self, Y):
        dm = self.dims()
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
        st = np.zeros((dm["d"], max(dm["d"] * dm["n"], 1)), order="F")
        ob = np.zeros(max(dm["r"], 1))
        self._chk(self.L.cora_problem_lambda_blocks(self.h, Y.ctypes.data_as(_dp), st.ctypes.data_as(_dp),
                                                    ob.ctypes.data_as(_dp)))
        return st[:, :dm["d"] * dm["n"]], ob[:dm["r"]]

    def tnt(self, x0, max_iterations=0, max_inner=0, grad_tol=0, pgrad_tol=0, max_seconds=0, verbose=False,
            host_stpcg=False):
        dm = self.dims()
        x0 = np.asfortranarray(np.asarray(x0, dtype=np.float64))
        assert x0.shape == (self.variable_size(), dm["rank"])
        opts = np.array([max_iterations, max_inner, grad_tol, pgrad_tol, max_seconds, float(verbose),
                         float(host_stpcg)])
        out = np.zeros_like(x0, order="F")
        st = np.zeros(7)
        self._chk(self.L.cora_problem_tnt(self.h, x0.ctypes.data_as(_dp), opts.ctypes.data_as(_dp),
                                          out.ctypes.data_as(_dp), st.ctypes.data_as(_dp)))
        return dict(x=out, f=st[0], grad_norm=st[1], pgrad_norm=st[2], iterations=int(st[3]), hvps=int(st[4]),
                    status=int(st[5]), seconds=st[6])

    def tnt_step(self, x, Delta, host_stpcg=False):
        """One outer TNT iteration from (x, Delta): cora_problem_tnt_step."""
        dm = self.dims()
        x = np.asfortranarray(np.asarray(x, dtype=np.float64))
        assert x.shape == (self.variable_size(), dm["rank"])
        out = np.zeros_like(x, order="F")
        st = np.zeros(8)
        self._chk(self.L.cora_problem_tnt_step(self.h, x.ctypes.data_as(_dp), C.c_double(Delta), int(host_stpcg),
                                               out.ctypes.data_as(_dp), st.ctypes.data_as(_dp)))
        return dict(x=out, f=st[0], Delta=st[1], inner=int(st[2]), rho=st[3], accepted=bool(st[4]), h_norm=st[5],
                    h_M_norm=st[6], status=int(st[7]))

    def certify(self, Y, eta, nx=10):
        dm = self.dims()
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
        out = np.zeros(3)
        x = np.zeros(dm["N"])
        self._chk(self.L.cora_problem_certify(self.h, Y.ctypes.data_as(_dp), C.c_double(eta), int(nx),
                                              out.ctypes.data_as(_dp), x.ctypes.data_as(_dp)))
        return dict(is_certified=bool(out[0]), theta=out[1], iters=int(out[2]), x=x)

    def certify_resident(self, Y, eta, nx=10, first=True):
        """The certification of solveCORA's own loop: the first level starts the eigensolver from the point, later levels
        from the block the previous call left on the device (cora_problem_certify_resident)."""
        dm = self.dims()
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
        out = np.zeros(3)
        x = np.zeros(dm["N"])
        self._chk(self.L.cora_problem_certify_resident(self.h, Y.ctypes.data_as(_dp), C.c_double(eta), int(nx), int(bool(first)),
                                                       out.ctypes.data_as(_dp), x.ctypes.data_as(_dp)))
        return dict(is_certified=bool(out[0]), theta=out[1], iters=int(out[2]), x=x)

    def saddle_escape(self, Y, theta, v, grad_tol=1e-4, pgrad_tol=1e-4):
        """saddleEscape (src/CORA.cpp:245-350) from the saddle point Y (N x (rank - 1); set_rank(rank) first, as
        solveCORA increments the rank before the call) along the certificate's direction v."""
        dm = self.dims()
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
        v = np.ascontiguousarray(np.asarray(v, dtype=np.float64))
        assert Y.shape == (self.variable_size(), dm["rank"] - 1) and v.shape == (self.variable_size(),)
        out = np.zeros((self.variable_size(), dm["rank"]), order="F")
        info = np.zeros(3)
        self._chk(self.L.cora_problem_saddle_escape(self.h, Y.ctypes.data_as(_dp), C.c_double(theta), v.ctypes.data_as(_dp),
                                                    C.c_double(grad_tol), C.c_double(pgrad_tol), out.ctypes.data_as(_dp),
                                                    info.ctypes.data_as(_dp)))
        return dict(x=out, f_saddle=info[0], f=info[1], moved=bool(info[2]))

    def project_solution(self, Y):
        """projectSolution (src/CORA.cpp:352-441): N x rank -> N x d."""
        dm = self.dims()
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
        assert Y.shape == (self.variable_size(), dm["rank"])
        out = np.zeros((self.variable_size(), dm["d"]), order="F")
        self._chk(self.L.cora_problem_project_solution(self.h, Y.ctypes.data_as(_dp), out.ctypes.data_as(_dp)))
        return out

    def certify_chain(self, Y, eta, nx=10, resident=False):
        """Two certifications in a row, the second started from the first one's Ritz block (host hand-over or resident)."""
        dm = self.dims()
        Y = np.asfortranarray(np.asarray(Y, dtype=np.float64))
        out = np.zeros(6)
        x = np.zeros(dm["N"])
        self._chk(self.L.cora_problem_certify_chain(self.h, Y.ctypes.data_as(_dp), C.c_double(eta), int(nx), int(bool(resident)),
                                                    out.ctypes.data_as(_dp), x.ctypes.data_as(_dp)))
        return dict(first=dict(is_certified=bool(out[0]), theta=out[1], iters=int(out[2])),
                    second=dict(is_certified=bool(out[3]), theta=out[4], iters=int(out[5]), x=x))

    def set_verification_lab(self, seed=True, ildl=True):
        """Test switches of certify()'s eigensolver stage (step 3 of fast_verification)."""
        self._chk(self.L.cora_problem_set_verification_lab(self.h, int(bool(seed)), int(bool(ildl))))

    def certification_reached_step3(self):
        v = C.c_int(0)
        self._chk(self.L.cora_problem_certification_reached_step3(self.h, C.byref(v)))
        return bool(v.value)

    def solve(self, x0, max_rank=10, verbose=False, max_seconds=0, max_iterations=0):
        dm = self.dims()
        x0 = np.asfortranarray(np.asarray(x0, dtype=np.float64))
        opts = np.array([max_iterations, 0, 0, 0, max_seconds, 0.0])
        out = np.zeros((self.variable_size(), dm["d"]), order="F")
        st = np.zeros(11)
        self._chk(self.L.cora_problem_solve(self.h, x0.ctypes.data_as(_dp), int(max_rank), int(verbose),
                                            opts.ctypes.data_as(_dp), out.ctypes.data_as(_dp),
                                            st.ctypes.data_as(_dp)))
        return dict(x=out, f=st[0], grad_norm=st[1], certified=bool(st[2]), eta=st[3], theta=st[4],
                    final_rank=int(st[5]), levels=int(st[6]), hvps=int(st[7]), seconds=st[8],
                    relaxation_certified=bool(st[9]), relaxation_rank=int(st[10]))

    def save_trajectory(self, X, path, g2o=False, robot=None):
        """saveSolnToTum / saveSolnToG20 for an N x d solution; robot = symbol character or None for all poses."""
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        self._chk(self.L.cora_problem_save_trajectory(self.h, X.ctypes.data_as(_dp), int(bool(g2o)),
                                                      ord(robot) if robot else 0, path.encode()))

    def precond_info(self):
        info = np.zeros(3)
        self._chk(self.L.cora_problem_precond_info(self.h, info.ctypes.data_as(_dp)))
        return dict(lam=info[0], nnz=int(info[1]), levels=int(info[2]))

    def cholesky_solve(self, B, m=None, shift=0.0, leaf_poses=16):
        dm = self.dims()
        m = dm["N"] - 1 if m is None else m
        B = np.asfortranarray(np.array(B, dtype=np.float64, copy=True))
        assert B.shape[0] == m
        info = (C.c_int64 * 3)()
        self._chk(self.L.cora_problem_cholesky_solve(self.h, int(m), C.c_double(shift), int(leaf_poses),
                                                     B.ctypes.data_as(_dp), B.shape[1], info))
        return B, dict(ok=bool(info[0]), nnz=int(info[1]), height=int(info[2]))

    def cholesky_probe(self, m=None, shift=0.0, leaf_poses=16, bump=None):
        """Host factorisation of (Q + shift I)[0:m, 0:m]: ok, nnz, first failing column, digest of L, negative direction.
        bump = {row: value}: added to those diagonal entries of a copy of Q first."""
        dm = self.dims()
        m = dm["N"] - 1 if m is None else m
        info = (C.c_int64 * 3)()
        digest = np.zeros(2)
        neg = np.zeros(dm["N"])
        rows = np.array(sorted(bump or {}), dtype=np.int32)
        vals = np.array([bump[r] for r in rows], dtype=np.float64) if len(rows) else np.zeros(0)
        self._chk(self.L.cora_problem_cholesky_probe_bumped(
            self.h, int(m), C.c_double(shift), int(leaf_poses), len(rows), rows.ctypes.data_as(C.POINTER(C.c_int32)),
            vals.ctypes.data_as(_dp), info, digest.ctypes.data_as(_dp), neg.ctypes.data_as(_dp)))
        return dict(ok=bool(info[0]), nnz=int(info[1]), failed_column=int(info[2]), digest=digest, negative_direction=neg)

    def plan_probe(self, shift, leaf_poses=2):
        """Host factorisation of (Q + shift I)[0:N-1] + the device solve plan built from it, no GPU involved."""
        info = (C.c_int64 * 4)()
        self._chk(self.L.cora_problem_plan_probe(self.h, C.c_double(shift), int(leaf_poses), info))
        return dict(stages=int(info[0]), nnzL=int(info[1]), blocks=int(info[2]), top_rows=int(info[3]))

    def context_ptr(self):
        c = self.L.cora_problem_context(self.h)
        if not c:
            raise HostError(self.L.cora_host_last_error().decode())
        return c


def manifold_op(kind, op, p, n, k=0, A=None, B=None, seed=0):
    """StiefelProduct(k, p, n) (kind "stiefel") / ObliqueManifold(p, n) (kind "oblique") of the reference, on the GPU."""
    L = _lib()
    cols = k * n if kind == "stiefel" else n
    f = lambda M: None if M is None else np.asfortranarray(np.asarray(M, dtype=np.float64))
    A, B = f(A), f(B)
    out = np.zeros(1) if op == "innerProduct" else np.zeros((p, cols), order="F")
    if L.cora_host_manifold_op(0 if kind == "stiefel" else 1, int(k), int(p), int(n), op.encode(),
                               None if A is None else A.ctypes.data_as(_dp), None if B is None else B.ctypes.data_as(_dp),
                               C.c_uint64(seed), out.ctypes.data_as(_dp)):
        raise HostError(L.cora_host_last_error().decode())
    return float(out[0]) if op == "innerProduct" else out


def block_cholesky_solve(A, block_sizes, B):
    """getBlockCholeskyFactorization + blockCholeskySolve of the reference's public interface, on the host."""
    import scipy.sparse as sp
    L = _lib()
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(A.indices, dtype=np.int32)
    va = np.ascontiguousarray(A.data, dtype=np.float64)
    bs = np.ascontiguousarray(block_sizes, dtype=np.int32)
    B = np.asfortranarray(np.asarray(B, dtype=np.float64).reshape(len(B), -1))
    X = np.zeros_like(B, order="F")
    ip = C.POINTER(C.c_int32)
    if L.cora_host_block_cholesky_solve(n, rp.ctypes.data_as(ip), ci.ctypes.data_as(ip), va.ctypes.data_as(_dp), len(bs),
                                        bs.ctypes.data_as(ip), B.shape[0], B.shape[1], B.ctypes.data_as(_dp),
                                        X.ctypes.data_as(_dp)):
        raise HostError(L.cora_host_last_error().decode())
    return X


def fast_verification(S, eta, X0=None, nx=1, max_iters=1000, lab=None, split=None):
    """CORA::fast_verification on an arbitrary symmetric scipy sparse matrix.  lab = dict(max_fill_factor, drop_tol,
    seed, ildl) exposes the knobs of step 3 (tests) and adds `step3` to the result."""
    import scipy.sparse as sp
    L = _lib()
    S = sp.csr_matrix(S)
    S.sort_indices()
    n = S.shape[0]
    rp = np.ascontiguousarray(S.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(S.indices, dtype=np.int32)
    va = np.ascontiguousarray(S.data, dtype=np.float64)
    out = np.zeros(3)
    x = np.zeros(n)
    xp = None
    if X0 is not None:
        X0 = np.asfortranarray(np.asarray(X0, dtype=np.float64).reshape(n, -1))
        nx = X0.shape[1]
        xp = X0.ctypes.data_as(_dp)
    if split is not None:
        rc = L.cora_host_fast_verification_pieces(n, rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), va.ctypes.data_as(_dp),
                                                  C.c_double(eta), xp, int(nx), int(split), int(max_iters),
                                                  out.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    elif lab is not None:
        out = np.zeros(4)
        opts = np.array([lab.get("max_fill_factor", 3.0), lab.get("drop_tol", 1e-3), float(lab.get("seed", True)),
                         float(lab.get("ildl", True))])
        rc = L.cora_host_fast_verification_lab(n, rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), va.ctypes.data_as(_dp),
                                               C.c_double(eta), xp, int(nx), int(max_iters), opts.ctypes.data_as(_dp),
                                               out.ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    else:
        rc = L.cora_host_fast_verification(n, rp.ctypes.data_as(_ip), ci.ctypes.data_as(_ip), va.ctypes.data_as(_dp),
                                           C.c_double(eta), xp, int(nx), int(max_iters), out.ctypes.data_as(_dp),
                                           x.ctypes.data_as(_dp))
    if rc:
        raise HostError(L.cora_host_last_error().decode())
    res = dict(is_certified=bool(out[0]), theta=out[1], iters=int(out[2]), x=x)
    if lab is not None:
        res["step3"] = bool(out[3])
    return res
