"""numpy restatement of the Riemannian truncated-Newton trust-region solver with a
Steihaug-Toint preconditioned CG inner loop -- TEST INFRASTRUCTURE ONLY.

The reference calls Optimization::Riemannian::TNT from the un-vendored submodule
david-m-rosen/Optimization (src/CORA.cpp:139-140; parameters :95-109; pinned
version unknown, sources absent), so this follows the published algorithm
(SE-Sync, Rosen et al. IJRR 2019, Alg. 3-4; Conn-Gould-Toint Alg. 7.5.1) on top
of the pinned oracle operators.  Parity of iterates with the reference is
UNPINNED (its tests/test_cora.cpp asserts nothing); this file pins the product's
C++ TNT (cora_amd/csrc/host/TNT.cpp) on converged values."""
import math

import numpy as np

from . import oracle as orc

DEFAULTS = dict(Delta0=5.0, eta1=0.05, eta2=0.9, alpha1=0.25, alpha2=3.0, max_TPCG_iterations=80,
                max_iterations=250, kappa_fgr=0.1, theta=0.8, preconditioned_gradient_tolerance=1e-6,
                gradient_tolerance=1e-6, Delta_tolerance=1e-5, relative_decrease_tolerance=1e-6,
                stepsize_tolerance=1e-6)


def stpcg(hess, precon, g, Delta, prm):
    s = np.zeros_like(g)
    r = g.copy()
    v = precon(r)
    p = -v
    r0 = math.sqrt(orc.inner(r, r))
    r_v = orc.inner(r, v)
    target = r0 * min(prm["kappa_fgr"], r0 ** prm["theta"])
    sig2, s_Mp, p_M2 = 0.0, 0.0, r_v
    it = 0
    while it < prm["max_TPCG_iterations"]:
        Hp = hess(p)
        it += 1
        kappa = orc.inner(p, Hp)
        alpha = r_v / kappa if kappa != 0 else float("inf")
        nxt = sig2 + 2 * alpha * s_Mp + alpha * alpha * p_M2
        if not (kappa > 0) or nxt >= Delta * Delta:
            tau = (-s_Mp + math.sqrt(s_Mp * s_Mp + p_M2 * (Delta * Delta - sig2))) / p_M2
            return s + tau * p, Delta, it
        s = s + alpha * p
        sig2 = nxt
        r = r + alpha * Hp
        v = precon(r)
        if math.sqrt(orc.inner(r, r)) <= target:
            break
        rv_new = orc.inner(r, v)
        beta = rv_new / r_v
        r_v = rv_new
        p = -v + beta * p
        s_Mp = beta * (s_Mp + alpha * p_M2)
        p_M2 = r_v + beta * beta * p_M2
    return s, math.sqrt(sig2), it


def tnt(Q, dm, x0, precond="jacobi", lam=None, perm=None, chol=None, **kw):
    """precond: "jacobi" | "none" | "chol" (RegularizedCholesky, src/CORA_problem.cpp:544-614, with
    the regularisation `lam` and the last translation pinned; `perm`: elimination order of the N - 1 rows -- the result
    of a solve does not depend on it, the fill does: natural order is hopeless beyond a few thousand poses)."""
    prm = dict(DEFAULTS)
    prm.update(kw)
    dinv = 1.0 / orc.diag(Q)
    if precond == "chol" and chol is None:   # (chol: a factor of the same matrix the caller already holds)
        import scipy.sparse as sp
        M = (Q.to_scipy() + lam * sp.eye(dm.N)).tocsr()[:dm.N - 1, :dm.N - 1]
        chol = orc.Cholesky(orc.CSR.from_scipy(M), perm=perm)
        assert chol.ok

    def precon_at(Y):
        if precond == "jacobi":
            return lambda V: orc.tangent_proj(dm, Y, V * dinv[:, None])
        if precond == "chol":
            return lambda V: chol.precond(dm, Y, V)
        return lambda V: orc.tangent_proj(dm, Y, V)

    x = np.asfortranarray(x0)
    G = orc.egrad(Q, x)
    f = 0.5 * orc.inner(x, G)
    grad = orc.tangent_proj(dm, x, G)
    P = precon_at(x)
    gn = math.sqrt(orc.inner(grad, grad))
    pgn = math.sqrt(orc.inner(P(grad), P(grad)))  # ||P g||: the measure saddleEscape uses (src/CORA.cpp:149)
    Delta = prm["Delta0"]
    hist = []
    status = "iteration_limit"
    hvps = 0
    last = dict(inner=0, rho=0.0, accepted=False, h_norm=0.0, h_M_norm=0.0, df=0.0, dmod=0.0)
    for it in range(prm["max_iterations"] + 1):
        hist.append((f, gn, pgn))
        if gn < prm["gradient_tolerance"]:
            status = "gradient"
            break
        if pgn < prm["preconditioned_gradient_tolerance"]:
            status = "preconditioned_gradient"
            break
        if it >= prm["max_iterations"]:
            break
        hess = lambda V, x=x, G=G: orc.hvp(Q, dm, x, G, V)  # noqa: E731
        h, hM, inner = stpcg(hess, P, grad, Delta, prm)
        hvps += inner + 1
        Hh = hess(h)
        dmod = -orc.inner(grad, h) - 0.5 * orc.inner(h, Hh)
        hn = math.sqrt(orc.inner(h, h))
        xp = orc.retract(dm, x, h)
        fp = orc.cost(Q, xp)
        df = f - fp
        rho = df / dmod if dmod != 0 else float("nan")
        rel = df / (math.sqrt(np.finfo(float).eps) + abs(f))
        accepted = (not math.isnan(rho)) and rho > prm["eta1"] and df > 0
        last = dict(inner=inner, rho=rho, accepted=accepted, h_norm=hn, h_M_norm=hM, df=df, dmod=dmod)
        if accepted:
            x = xp
            G = orc.egrad(Q, x)
            f = 0.5 * orc.inner(x, G)
            grad = orc.tangent_proj(dm, x, G)
            P = precon_at(x)
            gn = math.sqrt(orc.inner(grad, grad))
            pgn = math.sqrt(orc.inner(P(grad), P(grad)))  # ||P g||: the measure saddleEscape uses (src/CORA.cpp:149)
        if math.isnan(rho) or rho < prm["eta1"]:
            Delta = prm["alpha1"] * hM
        elif rho > prm["eta2"] and hM >= 0.99 * Delta:
            Delta = max(Delta, prm["alpha2"] * hM)
        if accepted and rel < prm["relative_decrease_tolerance"]:
            status = "relative_decrease"
            break
        if hn < prm["stepsize_tolerance"]:
            status = "stepsize"
            break
        if Delta < prm["Delta_tolerance"]:
            status = "trust_region"
            break
    return dict(x=x, f=f, grad_norm=gn, pgrad_norm=pgn, status=status, iterations=len(hist), hvps=hvps,
                history=hist, Delta=Delta, last=last)
