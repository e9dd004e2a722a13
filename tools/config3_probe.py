"""Developer probe: the config-3 staircase (10^4 poses from the odometry start) under different rank caps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
iters = int(os.environ.get("MAX_IT", "0"))
for cap in [int(a) for a in sys.argv[2:]] or [7, 10]:
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    x0 = P.op("getOdomInitialization")
    res = P.solve(x0, max_rank=cap, max_seconds=600, verbose=True, max_iterations=iters)
    print("cap %d:" % cap, {k: v for k, v in res.items() if k != "x"}, flush=True)
