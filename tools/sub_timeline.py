"""Timeline of the substitution blocks of the two sweeps inside an STPCG iteration (measurement build:
VARIANT_UNITS=kernels_tri_g0 tools/variant.sh "-DCORA_SUB_TIMES" python tools/sub_timeline.py [poses] [p])."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cora_amd import capi, host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_REGULARIZED_CHOLESKY)
P.update()
P.set_rank(p)
P.precond_info()
dm = P.dims()
h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
vecs = [h.dev_alloc(p) for _ in range(6)]
s, r, v, pk, hp, y = vecs
h.upload(np.random.default_rng(3).uniform(-1, 1, (dm["N"], p)), y)
h.project_to_manifold_dev(y, y)
h.set_point_dev(y)
grad = h.point_ptrs()[2]
for _ in range(3):
    done, step = h.stpcg_dev(grad, 1e30, s, r, v, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=12)
print("stpcg path", h.stpcg_path(), "iterations", done)
st = h.precond_stats() if hasattr(h, "precond_stats") else None
fp = h.L.cora_debug_sub_phases
fp.argtypes = [C.c_void_p, C.c_int]
nb = 4096
NP = 14
raw = np.zeros(2 * NP * nb, dtype=np.uint64)
assert fp(raw.ctypes.data, nb) == 0
for w, name in enumerate(("forward", "backward")):
    ph = raw[w * NP * nb:(w + 1) * NP * nb].reshape(nb, NP).astype(np.int64)
    ok = (ph[:, 0] > 0) & (ph[:, 3] > 0) & (ph[:, 5] > 0)
    ph = ph[ok]
    t0 = ph[:, 0].min()
    us = lambda x: x / 100.0   # wall_clock64 ticks at 100 MHz
    start, pro, lev, end = us(ph[:, 0] - t0), us(ph[:, 1] - ph[:, 0]), us(ph[:, 2] - ph[:, 1]), us(ph[:, 3] - ph[:, 2])
    wait, nlev = us(ph[:, 4]), ph[:, 5]
    pc = lambda x: "mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (x.mean(), *np.percentile(x, [10, 50, 90, 100]))
    print("%s sweep: %d solve blocks stamped, span %.2f us" % (name, len(ph), us(ph[:, 3].max() - t0)))
    print("  start            ", pc(start))
    print("  prologue (rhs->T)", pc(pro))
    print("  level loop       ", pc(lev), "| levels mean %.1f" % nlev.mean(), "| per level %.2f us" % (lev.sum() / nlev.sum()))
    print("    of it waiting for the level's entries (wave 0):", pc(wait), "| per level %.2f us" % (wait.sum() / nlev.sum()))
    print("  results (T->dst) ", pc(end))
    print("  whole block      ", pc(us(ph[:, 3] - ph[:, 0])))
    cyc = ph[:, 6:13].sum(axis=0) / max(nlev.sum(), 1)
    print("  shader-clock cycles per level (wave 0): wait %.0f | next level's request %.0f | tile reads + products %.0f | lane sums %.0f | "
          "barrier %.0f | rows into the tile %.0f | barrier %.0f | all %.0f" % (*cyc, cyc.sum()))
