"""Developer probe: the preconditioner apply alone (for rocprofv3 --kernel-trace)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                           precond=capi.PRECOND_REGULARIZED_CHOLESKY)
P.update(); P.set_rank(p); dm = P.dims()
print(P.precond_info())
L = capi.load(); h = C.c_void_p(P.context_ptr())
def alloc():
    q = C.POINTER(C.c_double)(); assert L.cora_dev_alloc(h, p, C.byref(q)) == 0; return q
x, y, o = alloc(), alloc(), alloc()
rng = np.random.default_rng(7); dp = C.POINTER(C.c_double)
Y = np.asfortranarray(rng.uniform(-1, 1, (dm["N"], p)))
assert L.cora_upload(h, Y.ctypes.data_as(dp), dm["N"], p, y) == 0
assert L.cora_project_to_manifold_dev(h, y, y) == 0
assert L.cora_set_point_dev(h, y) == 0
assert L.cora_upload(h, Y.ctypes.data_as(dp), dm["N"], p, x) == 0
ms = C.c_float()
for _ in range(3): L.cora_precondition_projected_dev(h, x, o)
L.cora_sync(h); L.cora_timer_start(h)
for _ in range(reps): L.cora_precondition_projected_dev(h, x, o)
L.cora_timer_stop_ms(h, C.byref(ms))
print("precond+proj %.1f us" % (ms.value * 1e3 / reps))
