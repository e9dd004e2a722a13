"""Full staircase on the 10^5-pose graph (BASELINE config 4 / 5 workload).
python tools/e2e_1e5.py [poses] [max seconds] [init: gt|odom] [verbose]
init gt  : the generator's ground truth (the point a front end with loop closures would hand over);
init odom: dead-reckoned odometry -- at 10^5 poses its drift (sigma_R sqrt(n) ~ 3 rad) leaves the basin so far
           behind that the reference's own TNT limits (250 iterations of at most 80 products, Delta0 = 5) end
           every level long before convergence."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from cora_amd import capi, host
from oracle import oracle as orc
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 240
init = sys.argv[3] if len(sys.argv) > 3 else "gt"
P, X_gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                                 precond=capi.PRECOND_REGULARIZED_CHOLESKY, ground_truth=True)
P.update()
x0 = P.op("getOdomInitialization") if init == "odom" else P.op("projectToManifold", X_gt)
dm = P.dims()
t0 = time.perf_counter()
P.precond_info()
t1 = time.perf_counter()
res = P.solve(x0, max_rank=7, max_seconds=secs, verbose=len(sys.argv) > 4)
print("n=%d init=%s: preconditioner %.2f s, staircase %.2f s, f=%.4f |g|=%.3e certified=%s theta=%.3e eta=%.3e levels=%d "
      "final rank %d hvps=%d (chi-square sized optimum ~ %d)"
      % (n, init, t1 - t0, res["seconds"], res["f"], res["grad_norm"], res["certified"], res["theta"], res["eta"],
         res["levels"], res["final_rank"], res["hvps"], n // 4))
