"""Developer stress run: random graphs x ranks x preconditioners x formulations, every operator against the
oracle.  python tools/stress.py [cases] [seed]"""
import os, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}
def note(k, v):
    worst[k] = max(worst.get(k, 0.0), float(v))

for it in range(cases):
    d = int(rng.choice([2, 3]))
    n = int(rng.choice([1, 2, 7, 65, 130, 700, 3000, 9000]))
    l = int(rng.choice([0, 1, 3, 12]))
    r = int(min(rng.choice([0, 1, n // 3, n]), n * l))
    loops = int(rng.choice([0, 0, 3, n // 10]))
    p = int(rng.choice([d, d + 1, d + 2, 7, 10]))
    implicit = bool(rng.integers(0, 2)) and (n + l >= 2)
    P = host.Problem.synthetic(dim=d, n_poses=n, n_landmarks=l, n_ranges=r, n_loops=loops, seed=int(rng.integers(1, 10**6)),
                               precond=capi.PRECOND_REGULARIZED_CHOLESKY)
    P.update()
    dm = P.dims()
    if dm["N"] < 2 or dm["nnz"] == 0:  # no measurement at all: Q = 0, nothing to regularise or solve
        continue
    _, _, rp, ci, va = P.matrix("DataMatrix")
    Q = orc.CSR(rp, ci, va, dm["N"]); dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    P.set_rank(p)
    tag = "d%d n%d l%d r%d loops%d p%d %s" % (d, n, l, r, loops, p, "implicit" if implicit else "explicit")
    try:
        if implicit:
            P.set_formulation(True)
            try:
                I = orc.Implicit(Q, dims)
            except AssertionError:  # reduced translation Laplacian singular (unobserved landmark): both must refuse
                try:
                    P.op("getRandomInitialGuess")
                    raise RuntimeError("implicit formulation accepted a singular translation block")
                except host.HostError:
                    print("ok (both refuse)", tag, flush=True)
                    continue
            try:
                Y = P.op("getRandomInitialGuess")
            except host.HostError as e:  # numerically singular block (e.g. an unobserved landmark is the pinned one)
                assert "not positive definite" in str(e)
                print("ok (refused: singular translation block)", tag, flush=True)
                continue
            G = I.product(Y); sc = max(np.abs(G).max(), 1e-9 * np.abs(va).max())  # Q_impl can vanish identically
            e_ = np.abs(P.op("Euclidean_gradient", Y) - G).max() / sc
            if e_ > 1e-8: print("  !! implicit egrad err %.2e" % e_, tag, flush=True)
            note("impl egrad", e_)
            V = P.op("tangent_space_projection", Y, rng.uniform(-1, 1, Y.shape))
            note("impl hvp", np.abs(P.op("Riemannian_Hessian_vector_product", Y, G, V) - I.hvp(Y, V)).max() / sc)
            note("impl cost", abs(P.op("evaluateObjective", Y) - I.cost(Y)) / max(abs(I.cost(Y)), 1e-9 * np.abs(va).max()))
            out = P.op("precondition", V)
            assert np.all(np.isfinite(out))
        else:
            Y = P.op("getRandomInitialGuess")
            note("manifold", np.abs(Y - orc.project_manifold(dims, Y)).max())
            G = orc.egrad(Q, Y); sc = max(np.abs(G).max(), 1e-300)
            note("egrad", np.abs(P.op("Euclidean_gradient", Y) - G).max() / sc)
            V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, Y.shape))
            note("hvp", np.abs(P.op("Riemannian_Hessian_vector_product", Y, G, V) - orc.hvp(Q, dims, Y, G, V)).max() / sc)
            note("retract", np.abs(P.op("retract", Y, 0.3 * V) - orc.retract(dims, Y, 0.3 * V)).max())
            lam = P.precond_info()["lam"]
            M = (Q.to_scipy() + lam * sp.eye(dims.N)).tocsr()[:dims.N - 1, :dims.N - 1]
            out = P.op("precondition", V)
            note("chol residual", np.abs(M @ out[:-1] - V[:-1]).max() / max(np.abs(V).max(), 1e-300))
            dg = Q.to_scipy().diagonal()
            if np.all(dg > 0):  # unobserved landmarks leave zeros on the diagonal: Jacobi is refused there
              P.set_preconditioner(capi.PRECOND_JACOBI)
              note("jacobi", np.abs(P.op("precondition", V) - V / dg[:, None]).max() / max(np.abs(V / dg[:, None]).max(), 1e-300))
            if n <= 700:
                P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY)
                P.set_rank(d)
                res = P.solve(P.op("getRandomInitialGuess"), max_rank=d + 3, max_seconds=20)
                note("solve cost", abs(orc.cost(Q, res["x"]) - res["f"]) / max(abs(res["f"]), 1.0))
    except Exception as e:
        print("FAILED", tag, "->", repr(e)[:300]); raise
    print("ok", tag, flush=True)
print({k: "%.2e" % v for k, v in worst.items()})
