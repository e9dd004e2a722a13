"""Developer probe: Hvp in the implicit formulation at the bench size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_REGULARIZED_CHOLESKY)
P.update(); P.set_rank(p)
t = time.time(); P.set_formulation(True); P.context_ptr(); print("implicit setup %.2fs" % (time.time() - t))
dm = P.dims()
h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
x, y, o = h.dev_alloc(p), h.dev_alloc(p), h.dev_alloc(p)
rng = np.random.default_rng(7)
Y = rng.uniform(-1, 1, (dm["N"], p)); Y[dm["d"] * dm["n"] + dm["r"]:] = 0
h.upload(Y, y); h.project_to_manifold_dev(y, y); h.set_point_dev(y)
h.upload(Y, x); h.tangent_space_projection_dev(x, x)
def timeit(name, fn, reps=100):
    for _ in range(5): fn()
    h.sync(); h.timer_start()
    for _ in range(reps): fn()
    print("%-30s %8.1f us" % (name, h.timer_stop_ms() * 1e3 / reps))
timeit("implicit hvp", lambda: h.hvp_dev(x, o))
timeit("implicit precond+proj", lambda: h.precondition_projected_dev(x, o))
P.set_formulation(False); P.context_ptr(); h.set_point_dev(y)
timeit("explicit hvp", lambda: h.hvp_dev(x, o))
timeit("explicit precond+proj", lambda: h.precondition_projected_dev(x, o))
