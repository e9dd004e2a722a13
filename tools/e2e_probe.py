"""End-to-end staircase timings: Plaza2 (reference data set), synthetic 10^4 poses (BASELINE config 3) and optionally
10^5 poses; prints seconds, Hessian-vector products, cost, certificate.  python tools/e2e_probe.py [plaza2|1e4|1e5 ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
which = sys.argv[1:] or ["plaza2", "1e4"]
for w in which:
    t0 = time.perf_counter()
    if w == "plaza2":
        P = host.Problem.from_pyfg(os.path.join(ROOT, "tests", "golden", "datasets", "plaza2.pyfg"))
        P.update()
        x0 = P.op("getRandomInitialGuess")
        max_rank = 10
    else:
        n = int(float(w))
        P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                                   precond=capi.PRECOND_REGULARIZED_CHOLESKY)
        P.update()
        x0 = P.op("getOdomInitialization")   # BASELINE config 3: odometry initialisation, staircase from r0 = 3
        max_rank = 7
    t1 = time.perf_counter()
    P.precond_info()
    t2 = time.perf_counter()
    res = P.solve(x0, max_rank=max_rank, max_seconds=600)
    t3 = time.perf_counter()
    print("%s: setup %.3f s, preconditioner %.3f s, staircase %.3f s (solver's clock %.3f) | f=%.6f certified=%s levels=%d "
          "final rank %d hvps=%d -> %.1f us per Hvp end to end" % (w, t1 - t0, t2 - t1, t3 - t2, res["seconds"], res["f"],
          res["certified"], res["levels"], res["final_rank"], res["hvps"], (t3 - t2) / max(res["hvps"], 1) * 1e6), flush=True)
