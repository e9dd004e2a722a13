"""End-to-end solve of a .pyfg file (examples/main.cpp's flow) with the host-side timing breakdown.
python tools/dataset_solve.py file.pyfg [max_rank]   (CORA_TRI_TIMING=1 for the breakdown)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
path = sys.argv[1]
max_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = time.perf_counter()
P = host.Problem.from_pyfg(path)
P.update()
P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY)
dm = P.dims()
t1 = time.perf_counter()
info = P.precond_info()
t2 = time.perf_counter()
x0 = P.op("getRandomInitialGuess")
res = P.solve(x0, max_rank=max_rank, max_seconds=600, verbose=bool(os.environ.get("CORA_TRI_TIMING")))
t3 = time.perf_counter()
print("%s: d=%d n=%d l=%d r=%d N=%d nnz=%d | parse+assemble %.3f s, preconditioner %.3f s (nnz(L)=%d), staircase %.3f s | "
      "f=%.6f |g|=%.2e certified=%s levels=%d final rank %d hvps=%d -> %.1f us per Hvp end to end"
      % (os.path.basename(path), dm["d"], dm["n"], dm["l"], dm["r"], dm["N"], dm["nnz"], t1 - t0, t2 - t1, info["nnz"], t3 - t2,
         res["f"], res["grad_norm"], res["certified"], res["levels"], res["final_rank"], res["hvps"],
         (t3 - t2) / max(res["hvps"], 1) * 1e6), flush=True)
