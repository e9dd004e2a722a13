"""CPU evidence for BASELINE config 3 (10^4-pose SE(3) chain + 5 000 ranges, odometry start): the Riemannian staircase of
src/CORA.cpp:26-243 run END TO END ON THE CPU ORACLE -- oracle/tnt.py (TNT + STPCG with the reference's limits,
src/CORA.cpp:95-109) on the pinned oracle operators, the oracle's sparse Cholesky as the RegularizedCholesky
preconditioner and as the PSD test of fast_verification (src/CORA_utils.cpp:36-51), a Lanczos eigenvector for the
direction of negative curvature, the reference's backtracking saddle escape (src/CORA.cpp:245-350).  No GPU, no
product code on the path (the C++ host only generates the graph, assembles Q and gives the odometry start).

It answers the round-2 review's question "where does the reference-equivalent CPU path stop from this start?":
every level ends on TNT's iteration limit far from stationarity, exactly like the GPU path.

python tools/oracle_staircase.py [poses] [max_rank] [threads] [outer iterations per level] [perturbation seed]   (10^4 poses: a few minutes per level on 8 cores)

With a perturbation seed s > 0 every entry of the start point is moved by one unit in its last place, up or down by a
coin drawn from numpy's default_rng(s) (round 6: how far the ORACLE's own end value moves under a last-digit change of its
start -- profiles/r06_config3_oracle_spread.txt)."""
import math, os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import host
from oracle import oracle as orc, staircase as ost, tnt as otnt

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000
max_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 7
orc.set_threads(int(sys.argv[3]) if len(sys.argv) > 3 else min(8, orc.max_threads()))
MAX_IT = int(sys.argv[4]) if len(sys.argv) > 4 else 250   # outer iterations per level (the reference: 250, src/CORA.cpp:97)

P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42)
P.update()
dm_ = P.dims()
_, _, rowptr, colidx, vals = P.matrix("DataMatrix")
Q = orc.CSR(rowptr, colidx, vals, dm_["N"])
dims = orc.Dims(dm_["d"], dm_["n"], dm_["r"], dm_["N"])
Qs = Q.to_scipy().tocsr()
x = orc.project_manifold(dims, P.op("getOdomInitialization"))
PERTURB = int(sys.argv[5]) if len(sys.argv) > 5 else 0
# certify_solution's singular-value shortcut (src/CORA_problem.cpp:1037-1049: extreme singular values of the point more than
# 1e6 apart count as certified).  Rounds 3-5 ran this tool WITHOUT it (argument 6 = 0 reproduces those runs); with it the tool
# takes every decision solveCORA takes.
SHORTCUT = (int(sys.argv[6]) != 0) if len(sys.argv) > 6 else True
# Which direction of negative curvature the escape takes.  The reference runs LOBPCG (source absent) until x'Sx < -eta/2 and
# takes whatever its iteration holds at that moment; any such vector is admissible.  "near": shift-invert Lanczos around
# -eta -- the admissible direction of SMALLEST curvature (theta ~ -0.1: what rounds 3-5 ran);  "min": shift-invert around a
# strongly negative shift -- the direction of LARGEST negative curvature, which is what an eigensolver for the minimum
# eigenvalue (the device's LOBPCG, and the reference's) converges towards (theta ~ -1 .. -30 here, alpha_0 = 100 tol / |theta|
# correspondingly shorter, src/CORA.cpp:287-288).
DIRECTION = sys.argv[7] if len(sys.argv) > 7 else "near"
if PERTURB > 0:
    coin = np.random.default_rng(PERTURB).integers(0, 2, size=x.shape).astype(bool)
    x = np.where(coin, np.nextafter(x, np.inf), np.nextafter(x, -np.inf))
    print("start perturbed by +-1 ulp per entry, seed %d" % PERTURB, flush=True)
# RegularizedCholesky: lambda = ||Q||_2 / (kappa_max - 1), kappa_max = 1e6 (src/CORA_problem.cpp:544-614)
lam_max = float(spla.eigsh(Qs, k=1, which="LA", tol=1e-3, return_eigenvectors=False)[0])
lam = lam_max / (1e6 - 1)


def pose_major_order(m):
    """Fill-reducing elimination order for a chain graph (per pose: its rotation rows, the range rows hanging off it,
    its translation; landmarks last), restricted to the first m rows."""
    d, n, r, N = dims.d, dims.n, dims.r, dims.N
    tb = d * n + r
    C = Qs[d * n:tb][:, tb:tb + n].tocsr()
    owner = np.full(r, -1, dtype=np.int64)
    has = np.diff(C.indptr) > 0
    owner[has] = C.indices[C.indptr[:-1][has]]
    key = np.empty(N, dtype=np.float64)
    for a in range(d):
        key[a:d * n:d] = np.arange(n) + 0.1 * a / d
    key[d * n:tb] = np.where(owner >= 0, owner + 0.5, n + 1.0)
    key[tb:tb + n] = np.arange(n) + 0.9
    key[tb + n:] = n + 2.0
    order = np.argsort(key, kind="stable")
    return order[order < m].astype(np.int32)


perm_full, perm_pin = pose_major_order(dims.N), pose_major_order(dims.N - 1)
print("N=%d nnz=%d f0=%.6e lambda_reg=%.6e" % (dims.N, dm_["nnz"], orc.cost(Q, x), lam), flush=True)
MIN_ETA, MAX_ETA, REL_ETA = 1e-7, 1e-1, 5e-6
t_start = time.time()
hvps = 0
rank = x.shape[1]
while rank <= max_rank:
    t0 = time.time()
    res = otnt.tnt(Q, dims, x, precond="chol", lam=lam, perm=perm_pin, max_iterations=MAX_IT)
    hvps += res["hvps"]
    x = res["x"]
    eta = min(max(res["f"] * REL_ETA, MIN_ETA), MAX_ETA)
    # certificate matrix S = Q - Lambda (src/CORA_problem.cpp:1105-1166), PSD test of S + eta I by factorisation
    Lst, lob = orc.lambda_blocks(Q, dims, x)
    d, nn, r = dims.d, dims.n, dims.r
    blocks = [sp.csr_matrix(Lst[:, i * d:(i + 1) * d]) for i in range(nn)]
    Lam = sp.block_diag(blocks + [sp.diags(lob)] + [sp.csr_matrix((dims.N - d * nn - r, dims.N - d * nn - r))], format="csr")
    S = (Qs - Lam).tocsr()
    M = (S + eta * sp.identity(dims.N)).tocsr()
    M.sort_indices()
    short = SHORTCUT and ost.rank_deficient(x)
    ok = short or orc.Cholesky(orc.CSR.from_scipy(M), perm=perm_full).ok
    sv = np.linalg.svd(x, compute_uv=False)
    print("rank %d: TNT %s after %d outer iterations, %d Hvps, f=%.6f |g|=%.3e |Pg|=%.3e  (%.0f s) -> eta=%.3g, sigma_max/sigma_min=%.3g, %s: %s"
          % (rank, res["status"], res["iterations"], res["hvps"], res["f"], res["grad_norm"], res["pgrad_norm"],
             time.time() - t0, eta, sv[0] / sv[-1] if sv[-1] > 0 else float("inf"),
             "certified by the singular-value shortcut" if short else "S + eta I PSD", ok), flush=True)
    if ok:
        break
    # direction of negative curvature (the reference: LOBPCG until x'Sx < -eta/2; any such vector serves the escape)
    # shift-invert Lanczos around a few negative shifts; the first Ritz pair below -eta / 2 is taken
    theta, v = None, None
    for shift in ((1e3, 1e4, 1e5, 1e2, 10.0) if DIRECTION == "min" else (eta, 10 * eta, 100 * eta, 1e3 * eta, 1e4 * eta, 1e5 * eta)):
        try:
            w, V = spla.eigsh(S.tocsc(), k=3, sigma=-shift, which="LM", tol=1e-6)
        except Exception as ex:  # singular shift: move on
            print("   shift %.3g: %s" % (-shift, type(ex).__name__), flush=True)
            continue
        k = int(np.argmin(w))
        if w[k] < -eta / 2:
            theta, v = float(w[k]), V[:, k]
            break
    if v is None:
        print("   no direction with x'Sx < -eta/2 found by the oracle's eigensolver: stop", flush=True)
        break
    rank += 1
    if rank > max_rank:
        break
    # saddleEscape, src/CORA.cpp:245-350
    Ya = np.hstack([x, np.zeros((dims.N, 1))])
    Yd = np.zeros_like(Ya)
    Yd[:, -1] = v
    FY = orc.cost(Q, Ya)
    chol = orc.Cholesky(orc.CSR.from_scipy(((Qs + lam * sp.identity(dims.N)).tocsr()[:dims.N - 1, :dims.N - 1]).tocsr()), perm=perm_pin)
    alpha = max(16 * 1e-6, 100 * 1e-4 / abs(theta))
    trials, best = [], None
    escaped = False
    while alpha >= 1e-6:
        Yt = orc.retract(dims, Ya, alpha * Yd)
        Ft = orc.cost(Q, Yt)
        trials.append((Ft, alpha))
        if Ft < FY:
            g = orc.rgrad(Q, dims, Yt)
            pg = chol.precond(dims, Yt, g)
            if math.sqrt(orc.inner(g, g)) > 1e-4 and math.sqrt(orc.inner(pg, pg)) > 1e-4:
                x, escaped = Yt, True
                break
        alpha /= 2
    if not escaped:
        Ft, a = min(trials)
        x = orc.retract(dims, Ya, a * Yd) if Ft < FY else Ya
    print("   theta=%.4e, saddle escape %s (f %.6f -> %.6f)" % (theta, "accepted" if escaped else "line search failed",
                                                                FY, orc.cost(Q, x)), flush=True)
print("staircase stopped at rank %d after %d Hvps, %.0f s: f=%.6f (chi-square sized optimum %d)"
      % (x.shape[1], hvps, time.time() - t_start, orc.cost(Q, x), n // 4), flush=True)
if x.shape[1] > dims.d:
    # projectSolution, src/CORA.cpp:352-441: the d dominant right singular directions, determinant fix, every pose block
    # to SO(d), range rows to the unit sphere -- then the refinement at rank d (src/CORA.cpp:198-233)
    d, nn, r = dims.d, dims.n, dims.r
    _, _, Vt = np.linalg.svd(x, full_matrices=False)
    Yd = x @ Vt[:d].T
    dets = np.array([np.linalg.det(Yd[i * d:(i + 1) * d]) for i in range(nn)])
    if (dets > 0).sum() < nn / 2:
        Yd[:, d - 1] = -Yd[:, d - 1]
    for i in range(nn):
        U, _, Wt = np.linalg.svd(Yd[i * d:(i + 1) * d])
        R = U @ Wt
        if np.linalg.det(R) < 0:
            U[:, -1] = -U[:, -1]
            R = U @ Wt
        Yd[i * d:(i + 1) * d] = R
    nr = np.linalg.norm(Yd[d * nn:d * nn + r], axis=1, keepdims=True)
    Yd[d * nn:d * nn + r] /= np.where(nr > 0, nr, 1.0)
    t0 = time.time()
    res = otnt.tnt(Q, dims, np.asfortranarray(Yd), precond="chol", lam=lam, perm=perm_pin, max_iterations=MAX_IT)
    print("rounded to rank %d: f=%.6f; refinement: TNT %s after %d outer iterations, %d Hvps, f=%.6f |g|=%.3e (%.0f s)"
          % (d, orc.cost(Q, Yd), res["status"], res["iterations"], res["hvps"], res["f"], res["grad_norm"], time.time() - t0),
          flush=True)

