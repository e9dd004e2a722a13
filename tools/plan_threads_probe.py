"""Preconditioner set-up under CORA_TRI_TIMING for a given CORA_TRI_THREADS (plan builder threads).  python tools/plan_threads_probe.py [poses]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_JACOBI)
P.update()
P.set_rank(5)
P.context_ptr()
for rep in range(2):
    t = time.perf_counter()
    P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY)
    P.precond_info()
    print("set-up %d: %.3f s" % (rep, time.perf_counter() - t), flush=True)
    P.set_preconditioner(capi.PRECOND_JACOBI)
