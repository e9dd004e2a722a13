"""BASELINE config 2: a reference data set at fixed rank, every operator of the inner loop on the GPU (resident
operands, HIP events) beside the CPU oracle (1 thread) in the same run.  python tools/dataset_probe.py file.pyfg [p]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
from oracle import oracle as orc
path = sys.argv[1]
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P = host.Problem.from_pyfg(path); P.update(); P.set_rank(p)
P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY); P.precond_info()
dm = P.dims(); _, _, rp, ci, va = P.matrix("DataMatrix")
Q = orc.CSR(rp, ci, va, dm["N"]); dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
rng = np.random.default_rng(7)
Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dm["N"], p)))
G = orc.egrad(Q, Y)
V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, (dm["N"], p)))
y, v, o = h.dev_alloc(p), h.dev_alloc(p), h.dev_alloc(p)
h.upload(Y, y); h.set_point_dev(y); h.upload(V, v)
orc.set_threads(1)
def gpu(fn, reps=300):
    for _ in range(10): fn()
    h.sync(); h.timer_start()
    for _ in range(reps): fn()
    return h.timer_stop_ms() * 1e3 / reps
def cpu(fn, budget=0.5):
    fn(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < budget: fn(); n += 1
    return (time.perf_counter() - t0) / n * 1e6
import scipy.sparse as sp
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fullsize import _pose_major_order   # fill-reducing order for the CPU factorisation
perm = _pose_major_order(Q, dims)
perm = perm[perm != dims.N - 1]
M = (Q.to_scipy() + P.precond_info()["lam"] * sp.eye(dims.N)).tocsr()[:dims.N - 1, :dims.N - 1]
M.sort_indices()
chol = orc.Cholesky(orc.CSR.from_scipy(M), perm=perm.astype(np.int32))
assert chol.ok
rows = [
    ("Q X (dataMatrixProduct)", gpu(lambda: h.spmm_dev(v, p, o)), cpu(lambda: orc.spmm(Q, V, rowwise=True))),
    ("Riemannian Hvp", gpu(lambda: h.hvp_dev(v, o)), cpu(lambda: orc.hvp(Q, dims, Y, G, V))),
    ("tangent projection", gpu(lambda: h.tangent_space_projection_dev(v, o)), cpu(lambda: orc.tangent_proj(dims, Y, V))),
    ("retraction", gpu(lambda: h.retract_dev(v, 1.0, o)), cpu(lambda: orc.retract(dims, Y, V))),
    ("Cholesky preconditioner + projection", gpu(lambda: h.precondition_projected_dev(v, o)), cpu(lambda: chol.precond(dims, Y, V))),
]
print("%s: d=%d n=%d l=%d r=%d N=%d nnz=%d, p=%d" % (os.path.basename(path), dm["d"], dm["n"], dm["l"], dm["r"], dm["N"], dm["nnz"], p))
print("| operator | MI355X us | CPU oracle, 1 thread us | ratio |\n|---|---|---|---|")
for name, g, c in rows:
    print("| %s | %.1f | %.0f | %.0fx |" % (name, g, c, c / g))
