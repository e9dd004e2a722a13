"""Full staircase twice in one process (fresh Problem each time) with CORA_TRACE_BITS=1: the [bits] lines of the two
runs must be identical.  python tools/determinism_solve.py [poses] [init: odom|gt] [repeats]"""
import os, sys
os.environ["CORA_TRACE_BITS"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000
init = sys.argv[2] if len(sys.argv) > 2 else "odom"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    P, X_gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                                     precond=capi.PRECOND_REGULARIZED_CHOLESKY, ground_truth=True)
    P.update()
    x0 = P.op("getOdomInitialization") if init == "odom" else P.op("projectToManifold", X_gt)
    print("run %d" % rep, flush=True)
    res = P.solve(x0, max_rank=7, max_seconds=300)
    print("result f=%s |g|=%s certified=%s levels=%d rank=%d hvps=%d %.2fs" % (
        float(res["f"]).hex(), float(res["grad_norm"]).hex(), res["certified"], res["levels"], res["final_rank"],
        res["hvps"], res["seconds"]), flush=True)
