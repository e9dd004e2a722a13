"""Phase timing of the 10^5-pose staircase from the ground truth (what bench.py's staircase_from_ground_truth runs),
with the host's CORA_TRI_TIMING ticks on stderr.   python tools/staircase_timing.py [poses]"""
import os, sys, time
os.environ["CORA_TRI_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
for rep in range(2):
    t0 = time.time()
    P, x_gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, ground_truth=True,
                                     precond=capi.PRECOND_REGULARIZED_CHOLESKY if len(sys.argv) > 2 else capi.PRECOND_JACOBI)
    t1 = time.time()
    P.update()
    t2 = time.time()
    x = P.op("projectToManifold", x_gt)
    t3 = time.time()
    res = P.solve(x, max_rank=7, max_seconds=120, verbose=len(sys.argv) > 2)
    t4 = time.time()
    print("rep %d: generate %.3f update %.3f project %.3f solve %.3f (solver %.3f) hvps %d levels %d" %
          (rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, res["seconds"], res["hvps"], res["levels"]), flush=True)
