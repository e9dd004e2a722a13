import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from cora_amd import capi, host
from oracle import oracle as orc
kw = dict(dim=3, n_poses=150, n_landmarks=3, n_ranges=120, n_loops=6, seed=11, precond=capi.PRECOND_REGULARIZED_CHOLESKY)
def mk(imp):
    P = host.Problem.synthetic(**kw); P.update(); P.set_formulation(imp); P.set_rank(5); return P
Pe, Pi = mk(False), mk(True)
dm = Pi.dims(); _, _, rp, ci, va = Pi.matrix("DataMatrix")
Q = orc.CSR(rp, ci, va, dm["N"]); dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"]); I = orc.Implicit(Q, dims)
x0 = Pe.op("getRandomInitialGuess")
for name, P, x in (("expl", Pe, x0), ("impl", Pi, np.asfortranarray(x0[:I.dm.N]))):
    r = P.tnt(x, grad_tol=1e-7, pgrad_tol=1e-7)
    print(name, {k: v for k, v in r.items() if k != "x"})
    r = P.tnt(r["x"], grad_tol=1e-7, pgrad_tol=1e-7)
    print(name, "again", {k: v for k, v in r.items() if k != "x"})
print("f0", orc.cost(Q, x0), I.cost(x0[:I.dm.N]))
