"""Developer probe: per-operator device time of the solver loop at a given size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                           precond=capi.PRECOND_REGULARIZED_CHOLESKY)
P.update(); P.set_rank(p); dm = P.dims()
t = time.time(); info = P.precond_info(); print("precond setup %.2fs" % (time.time() - t), info)
import ctypes as C
st = (C.c_int64 * 4)(); capi.load().cora_precond_stats(C.c_void_p(P.context_ptr()), st); print("solve stages", st[0], "nnz(W)", st[1], "nnz(L)", st[2], "top rows", st[3])
import ctypes as C
L = capi.load()
h = C.c_void_p(P.context_ptr())
def alloc():
    q = C.POINTER(C.c_double)(); assert L.cora_dev_alloc(h, p, C.byref(q)) == 0; return q
x, y, o = alloc(), alloc(), alloc()
rng = np.random.default_rng(7)
Y = np.asfortranarray(rng.uniform(-1, 1, (dm["N"], p)))
dp = C.POINTER(C.c_double)
assert L.cora_upload(h, Y.ctypes.data_as(dp), dm["N"], p, y) == 0
assert L.cora_project_to_manifold_dev(h, y, y) == 0
assert L.cora_set_point_dev(h, y) == 0
V = np.asfortranarray(rng.uniform(-1, 1, (dm["N"], p)))
assert L.cora_upload(h, V.ctypes.data_as(dp), dm["N"], p, x) == 0
assert L.cora_tangent_space_projection_dev(h, x, x) == 0
ms = C.c_float(); val = C.c_double()
def timeit(name, fn, reps=50):
    for _ in range(3): fn()
    L.cora_sync(h); L.cora_timer_start(h)
    for _ in range(reps): fn()
    L.cora_timer_stop_ms(h, C.byref(ms))
    print("%-28s %9.1f us" % (name, ms.value * 1e3 / reps))
timeit("hvp", lambda: L.cora_hvp_dev(h, x, o), 200)
timeit("precond+proj (cholesky)", lambda: L.cora_precondition_projected_dev(h, x, o))
L.cora_precond_setup(h, capi.PRECOND_JACOBI)
timeit("precond+proj (jacobi)", lambda: L.cora_precondition_projected_dev(h, x, o), 200)
timeit("tangent projection", lambda: L.cora_tangent_space_projection_dev(h, x, o), 200)
timeit("axpby", lambda: L.cora_axpby_dev(h, C.c_double(0.5), x, C.c_double(1.0), o), 200)
timeit("dot (incl. sync)", lambda: L.cora_dot_dev(h, x, o, p, C.byref(val)), 200)
timeit("retract", lambda: L.cora_retract_dev(h, x, C.c_double(1.0), o), 200)
timeit("set_point (incl. sync)", lambda: L.cora_set_point_dev(h, y), 50)
timeit("objective (incl. sync)", lambda: L.cora_objective_dev(h, y, C.byref(val)), 50)
