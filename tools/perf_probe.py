"""Developer probe (not the bench): times hvp_dev / spmm_dev on a synthetic chain
graph built with the test generator.  Usage: python tools/perf_probe.py [n] [p]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cora_amd import capi  # noqa: E402
from synth import make_problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t0 = time.time()
A, Q, dm = make_problem(d=3, n=n, n_landmarks=10, n_ranges=n // 2, seed=42)
print("build %.1fs N=%d nnz=%d" % (time.time() - t0, dm.N, Q.nnz), flush=True)
c = capi.Context(dm.d, dm.n, dm.r, dm.n_trans, Q.rowptr, Q.col, Q.val)
c.set_rank(p)
print(c.format_stats())
rng = np.random.default_rng(7)
Y = rng.uniform(-1, 1, (dm.N, p))
y = c.dev_alloc(p); x = c.dev_alloc(p); o = c.dev_alloc(p)
c.upload(Y, y)
c.project_to_manifold_dev(y, y)
c.set_point_dev(y)
c.upload(rng.uniform(-1, 1, (dm.N, p)), x)
c.tangent_space_projection_dev(x, x)
b_spmm = 12 * Q.nnz + 4 * (dm.N + 1) + 16 * dm.N * p
b_hvp = b_spmm + 8 * (dm.dn + dm.r) * p + 8 * (dm.n * dm.d ** 2 + dm.r)
for name, fn, by in (("spmm", lambda: c.spmm_dev(x, p, o), b_spmm), ("hvp", lambda: c.hvp_dev(x, o), b_hvp)):
    for _ in range(20):
        fn()
    c.sync()
    reps = 200
    c.timer_start()
    for _ in range(reps):
        fn()
    ms = c.timer_stop_ms()
    us = ms * 1e3 / reps
    print("%s: %.2f us  %.2f MB algorithmic  %.0f GB/s (%.1f%% of 8 TB/s)" % (name, us, by / 1e6, by / us / 1e3, by / us / 1e3 / 80))

# PCIe-inclusive rate of the host-pointer entry point (3 uploads + 1 download per call)
import time as _t
G = c.Euclidean_gradient(Y)
Vh = rng.uniform(-1, 1, (dm.N, p))
c.Riemannian_Hessian_vector_product(Y, G, Vh)
t0 = _t.perf_counter()
for _ in range(5):
    c.Riemannian_Hessian_vector_product(Y, G, Vh)
dt = (_t.perf_counter() - t0) / 5
print("host-pointer Hvp (PCIe inclusive): %.2f ms per call = %.0f Hvp/s" % (dt * 1e3, 1 / dt))
