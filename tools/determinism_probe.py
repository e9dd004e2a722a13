"""Where does run-to-run nondeterminism enter?  Runs every layer twice on the same input and compares BITS.
python tools/determinism_probe.py [poses] [rank]
Layers: preconditioner set-up (lambda), single operators, one STPCG solve, TNT truncated at K outer iterations."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def h(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]


def make():
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42,
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    P.set_rank(rank)
    return P


P1, P2 = make(), make()
x0 = P1.op("getOdomInitialization")
print("x0", h(x0), h(P2.op("getOdomInitialization")))
i1, i2 = P1.precond_info(), P2.precond_info()
print("lambda", i1["lam"].hex(), i2["lam"].hex(), "nnz", i1["nnz"], i2["nnz"])
if x0.shape[1] < rank:
    x0 = np.hstack([x0, np.zeros((x0.shape[0], rank - x0.shape[1]))])
Y = P1.op("projectToManifold", x0)
print("project", h(Y), h(P2.op("projectToManifold", x0)), h(P1.op("projectToManifold", x0)))
G = P1.op("Euclidean_gradient", Y)
print("egrad", h(G), h(P2.op("Euclidean_gradient", Y)), h(P1.op("Euclidean_gradient", Y)))
rng = np.random.default_rng(1)
V = P1.op("tangent_space_projection", Y, rng.uniform(-1, 1, Y.shape))
for name, args in (("Riemannian_Hessian_vector_product", (Y, G, V)), ("precondition", (Y, V)),
                   ("Riemannian_gradient", (Y,))):
    try:
        outs = [h(P.op(name, *args)) for P in (P1, P2, P1, P2, P1)]
        print(name, outs, "SAME" if len(set(outs)) == 1 else "DIFFER")
    except Exception as e:  # noqa
        print(name, "skipped:", e)
for env in ({}, {"CORA_NO_SWEEP_FUSE": "1"}, {"CORA_NO_FUSE": "1"}):
    for k, v in env.items():
        os.environ[k] = v
    print("env", env, flush=True)
    first_bad = None
    for K in (1, 2, 3, 5, 8, 12, 20, 40, 80):
        rs = [P.tnt(Y, max_iterations=K) for P in (P1, P2, P1)]
        hs = [h(r["x"]) for r in rs]
        same = len(set(hs)) == 1
        print("  TNT K=%3d f=%s hvps=%s x=%s %s" % (K, [r["f"].hex() for r in rs], [r["hvps"] for r in rs], hs,
                                                     "SAME" if same else "DIFFER"), flush=True)
        if not same and first_bad is None:
            first_bad = K
            break
    for k in env:
        del os.environ[k]
rs = [P.tnt(Y, max_iterations=40, host_stpcg=True) for P in (P1, P2, P1)]
print("host-driven STPCG K=40", [r["f"].hex() for r in rs], [r["hvps"] for r in rs], [h(r["x"]) for r in rs])
