"""Stand-alone preconditioner apply and STPCG iteration, microseconds (host events).  python tools/apply_probe.py [poses] [p]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P, x_gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                                 precond=capi.PRECOND_REGULARIZED_CHOLESKY, ground_truth=True)
P.update(); P.set_rank(p); dm = P.dims()
P.precond_info()
h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
s, r, z, pk, hp, y = [h.dev_alloc(p) for _ in range(6)]
Yh = np.zeros((dm["N"], p)); Yh[:, :dm["d"]] = x_gt
h.upload(Yh, y); h.project_to_manifold_dev(y, y); h.set_point_dev(y)
h.upload(np.random.default_rng(0).standard_normal((dm["N"], p)), r)
for _ in range(5): h.precondition_projected_dev(r, z)
h.sync(); h.timer_start()
for _ in range(100): h.precondition_projected_dev(r, z)
print("apply %.1f us" % (h.timer_stop_ms() * 10))
grad = h.point_ptrs()[2]
h.stpcg_dev(grad, 1e30, s, r, z, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=8); h.sync()
t0 = time.perf_counter()
done, _ = h.stpcg_dev(grad, 1e30, s, r, z, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=60); h.sync()
print("stpcg path %d: %.1f us per iteration" % (h.stpcg_path(), (time.perf_counter() - t0) / max(done, 1) * 1e6))
