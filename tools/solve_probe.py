"""Developer probe: TNT with the RegularizedCholesky preconditioner on a synthetic graph."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = capi.PRECOND_JACOBI if (len(sys.argv) > 3 and sys.argv[3] == "jacobi") else capi.PRECOND_REGULARIZED_CHOLESKY
t = time.time(); P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=kind)
P.update(); P.set_rank(p); print("build %.1fs" % (time.time() - t), P.dims())
x0 = P.op("getOdomInitialization" if (len(sys.argv) > 4 and sys.argv[4] == "odom") else "getRandomInitialGuess")
t = time.time(); info = P.precond_info(); print("precond setup %.2fs" % (time.time() - t), info)
f0 = P.op("evaluateObjective", x0)
t = time.time(); r = P.tnt(x0, max_seconds=300); dt = time.time() - t
if len(sys.argv) > 5 and sys.argv[5] == "solve":
    P.set_rank(3); x0 = P.op("getOdomInitialization"); s = P.solve(x0, max_rank=8, max_seconds=300); print("solveCORA:", {k: v for k, v in s.items() if k != "x"})
print("f0 %.4e -> f %.6e |g| %.2e iters %d hvps %d status %d  %.3fs (%.1f us per Hvp-iteration)" % (
    f0, r["f"], r["grad_norm"], r["iterations"], r["hvps"], r["status"], dt, dt / max(r["hvps"], 1) * 1e6))
