"""What the solve plan of a reference data set looks like: python tools/dataset_plan.py name [name ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
for name in sys.argv[1:]:
    P = host.Problem.from_pyfg(os.path.join(ROOT, "tests", "golden", "datasets", name + ".pyfg"))
    P.update(); P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY); P.set_rank(P.dims()["d"] + 1)
    info = P.precond_info(); dm = P.dims()
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    print(name, "N", dm["N"], "nnz(Q)", dm["nnz"], {k: info[k] for k in info if k in ("nnz", "stages", "lam")}, h.precond_entries(), flush=True)
