"""Small synthetic RA-SLAM graphs for tests (numpy; the product generator is the
C++ host).  Same recipe as SURVEY 8(d): SE(d) random-walk odometry chain,
landmarks, pose->landmark ranges; optional loop closures to break bandedness."""
import numpy as np

from oracle import assemble as asm


def _rot(d, rng, sigma):
    if d == 2:
        return asm._from_angle(rng.normal(0, sigma))
    w = rng.normal(0, sigma, 3)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_graph(d=3, n=200, n_landmarks=3, n_ranges=100, n_loops=0, seed=42):
    rng = np.random.default_rng(seed)
    g = asm.PyFG()
    g.dim = d
    R = [np.eye(d)]
    t = [np.zeros(d)]
    for i in range(n):
        g.poses["A%d" % i] = i
    e1 = np.zeros(d)
    e1[0] = 1.0
    cov_rpm = np.diag([0.05 ** 2] * d + [0.01 ** 2] * (3 if d == 3 else 1))
    for i in range(n - 1):
        dR = _rot(d, rng, 0.05)
        dt = e1 + rng.normal(0, 0.1, d)
        R.append(R[-1] @ dR)
        t.append(t[-1] + R[-2] @ dt)
        Rm = dR @ _rot(d, rng, 0.01)
        tm = dt + rng.normal(0, 0.05, d)
        g.rpms.append(("A%d" % i, "A%d" % (i + 1), Rm, tm, cov_rpm))
    seen = set()
    for _ in range(n_loops):
        i, j = sorted(rng.integers(0, n, 2))
        if j - i < 2 or (i, j) in seen:
            continue
        seen.add((i, j))
        Rm = R[i].T @ R[j] @ _rot(d, rng, 0.01)
        tm = R[i].T @ (t[j] - t[i]) + rng.normal(0, 0.05, d)
        g.rpms.append(("A%d" % i, "A%d" % j, Rm, tm, cov_rpm))
    T = np.array(t)
    lo, hi = T.min(0) - 20, T.max(0) + 20
    L = rng.uniform(lo, hi, (n_landmarks, d))
    for k in range(n_landmarks):
        g.landmarks["L%d" % k] = k
    used = set()
    while len(g.ranges) < n_ranges and n_landmarks > 0:
        i, k = int(rng.integers(0, n)), int(rng.integers(0, n_landmarks))
        if (i, k) in used:
            continue
        used.add((i, k))
        dist = np.linalg.norm(T[i] - L[k]) + rng.normal(0, 0.1)
        g.ranges.append(("A%d" % i, "L%d" % k, abs(dist), 0.01))
    return g


def make_problem(**kw):
    from oracle import oracle as orc
    g = make_graph(**kw)
    A = asm.assemble(g)
    Q = orc.CSR.from_scipy(A["Q"])
    dm = orc.Dims(A["d"], A["n"], A["r"], A["N"])
    return A, Q, dm
