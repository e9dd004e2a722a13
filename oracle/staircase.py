"""numpy restatement of the steps of the Riemannian staircase AROUND the trust-region solves -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and tools/ as the checker; the product never imports it).

  saddle_escape    src/CORA.cpp:245-350   backtracking line search from a saddle point along e_{r+1} v'
  project_solution src/CORA.cpp:352-441   rank-d truncated SVD, orientation fix, SO(d) / unit-sphere projection
  cert_eta         src/CORA.cpp:111-116,152-154   eta = clamp(f * 5e-6, 1e-7, 1e-1)
  rank_deficient   src/CORA_problem.cpp:1037-1049 the singular-value shortcut of certify_solution

on the pinned oracle operators (oracle/oracle.py; each function cites the lines it follows).  Together with oracle/tnt.py
(the trust-region solver, sources of the reference absent) and the oracle's sparse Cholesky (the PSD test of
fast_verification, src/CORA_utils.cpp:36-51) this is every decision solveCORA takes between two TNT calls, so the GPU
staircase can be checked LEVEL BY LEVEL from its own points (tests/test_gpu_staircase.py)."""
import math

import numpy as np

from . import oracle as orc

MIN_CERT_ETA, MAX_CERT_ETA, REL_CERT_ETA = 1e-7, 1e-1, 5e-6  # src/CORA.cpp:111-113


def cert_eta(f):
    """thresholdVal(f * REL_CERT_ETA, MIN_CERT_ETA, MAX_CERT_ETA), src/CORA.cpp:152-154."""
    return min(max(f * REL_CERT_ETA, MIN_CERT_ETA), MAX_CERT_ETA)


def rank_deficient(Y):
    """src/CORA_problem.cpp:1037-1049: certify_solution treats a point whose extreme singular values differ by more than
    1e6 as certified without looking at S (a level the staircase entered along a short escape step and TNT left at once)."""
    sv = np.linalg.svd(np.asarray(Y), compute_uv=False)
    return bool(sv[-1] == 0.0 or sv[0] / sv[-1] > 1e6)  # (an exactly zero column: the ratio is infinite)


def saddle_escape(Q, dims, precond, Y, theta, v, gradient_tolerance=1e-4, preconditioned_gradient_tolerance=1e-4):
    """src/CORA.cpp:245-350.  Y: N x (r - 1) saddle point, v: direction of negative curvature of Q - Lambda(Y), theta its
    curvature.  precond(Ytest, V) -> tangent_space_projection(Ytest, precondition(V)) (the caller binds the factor).
    Returns (Y_next N x r, info): info["accepted"] -- a trial point met all three conditions (:307-311);
    info["fallback"] -- none did and the trial point of least cost was taken (:321-333); info["trials"] = [(alpha, f)]."""
    N, r = Y.shape[0], Y.shape[1] + 1
    Y_aug = np.zeros((N, r), order="F")
    Y_aug[:, :r - 1] = Y                      # :272-273
    FY = orc.cost(Q, Y_aug)                   # :276
    Ydot = np.zeros((N, r), order="F")
    Ydot[:, r - 1] = v                        # :278-279
    alpha_min = 1e-6                          # :286
    alpha = max(16 * alpha_min, 100 * gradient_tolerance / abs(theta))  # :287-288
    trials = []
    while alpha >= alpha_min:                 # :296
        Yt = orc.retract(dims, Y_aug, alpha * Ydot)  # :298
        Ft = orc.cost(Q, Yt)                  # :304
        trials.append((alpha, Ft))
        if Ft < FY:                           # :315 (the gradient norms only matter when the cost decreased)
            g = orc.rgrad(Q, dims, Yt)        # :305-306
            pg = precond(Yt, g)               # :307-310
            if math.sqrt(orc.inner(g, g)) > gradient_tolerance and math.sqrt(orc.inner(pg, pg)) > preconditioned_gradient_tolerance:
                return Yt, dict(accepted=True, fallback=False, alpha=alpha, f_saddle=FY, f=Ft, trials=trials)
        alpha /= 2                            # :320
    a_min, f_min = min(trials, key=lambda t: t[1])  # :330-334 (std::min_element: the first of equal minima)
    if f_min < FY:                            # :336-339
        Yt = orc.retract(dims, Y_aug, a_min * Ydot)
        return Yt, dict(accepted=False, fallback=True, alpha=a_min, f_saddle=FY, f=f_min, trials=trials)
    return Y_aug, dict(accepted=False, fallback=False, alpha=0.0, f_saddle=FY, f=FY, trials=trials)  # :340-348


def project_to_SOd(M):
    """src/CORA_utils.cpp:188-203 (projectToSOd): U V' when det U det V > 0, else the last column of U negated."""
    U, _, Vt = np.linalg.svd(M)
    if np.linalg.det(U) * np.linalg.det(Vt) > 0:   # :192-196
        return U @ Vt
    U = U.copy()
    U[:, -1] *= -1                                  # :198-200
    return U @ Vt


def project_solution(dims, Y):
    """src/CORA.cpp:352-441.  Returns the N x d rounded point.  The thin SVD fixes U_d Sigma_d only up to the signs of its
    columns; what follows (orientation fix, per-block projection) commutes with a right multiplication by a rotation,
    so two correct implementations agree up to ONE d x d rotation on the right: compare costs and Gram matrices."""
    d, n, r = dims.d, dims.n, dims.r
    U, sig, _ = np.linalg.svd(Y, full_matrices=False)        # :361
    Yd = U[:, :d] * sig[:d]                                   # :363-378
    dets = np.array([np.linalg.det(Yd[i * d:(i + 1) * d]) for i in range(n)])  # :383-393
    if n > 0 and (dets > 0).sum() < n // 2:                   # :402 (integer n / 2)
        Yd[:, d - 1] = -Yd[:, d - 1]                          # :408-411
    for i in range(n):                                        # :415-417
        Yd[i * d:(i + 1) * d] = project_to_SOd(Yd[i * d:(i + 1) * d])
    rot = d * n
    nr = np.linalg.norm(Yd[rot:rot + r], axis=1, keepdims=True)   # :421-422
    Yd[rot:rot + r] /= np.where(nr > 0, nr, 1.0)
    return np.asfortranarray(Yd)
