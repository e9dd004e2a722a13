"""Developer probe: the device STPCG loop alone (for rocprofv3 --kernel-trace; see tools/stpcg_trace.sh).

    python tools/stpcg_probe.py [n_poses] [p] [iterations] [dim]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
its = int(sys.argv[3]) if len(sys.argv) > 3 else 40
d = int(sys.argv[4]) if len(sys.argv) > 4 else 3
P, x_gt = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                                 precond=capi.PRECOND_REGULARIZED_CHOLESKY, ground_truth=True)
P.update(); P.set_rank(p); dm = P.dims()
print(P.precond_info())
h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
s, r, z, pk, hp, y = [h.dev_alloc(p) for _ in range(6)]
Yh = np.zeros((dm["N"], p))  # at the ground truth the Hessian is PSD up to the noise: CG is not cut short
Yh[:, :dm["d"]] = x_gt
h.upload(Yh, y)
h.project_to_manifold_dev(y, y)
h.set_point_dev(y)
grad = h.point_ptrs()[2]
h.stpcg_dev(grad, 1e30, s, r, z, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=8)
h.sync()
for rep in range(int(os.environ.get("PROBE_REPEATS", "1"))):
    t0 = time.perf_counter()
    done, _ = h.stpcg_dev(grad, 1e30, s, r, z, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=its)
    h.sync()
    print("stpcg path %d: %d iterations, %.1f us each" % (h.stpcg_path(), done, (time.perf_counter() - t0) / max(done, 1) * 1e6))
