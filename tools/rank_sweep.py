"""Per-rank roofline sweep (BASELINE configs 3 and 5): times Q*X, the Hvp and the certificate
operator for several ranks on the synthetic graphs and prints a markdown table."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host

def run(n, ranks):
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42)
    P.update(); dm = P.dims()
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    c = capi.Context(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], rowptr, colidx, vals)
    rng = np.random.default_rng(7)
    for p in ranks:
        c.set_rank(p)
        y, x, o = c.dev_alloc(p), c.dev_alloc(p), c.dev_alloc(p)
        c.upload(rng.uniform(-1, 1, (dm["N"], p)), y); c.project_to_manifold_dev(y, y); c.set_point_dev(y)
        c.upload(rng.uniform(-1, 1, (dm["N"], p)), x); c.tangent_space_projection_dev(x, x)
        b_spmm = 12 * dm["nnz"] + 4 * (dm["N"] + 1) + 16 * dm["N"] * p
        b_hvp = b_spmm + 8 * (dm["d"] * dm["n"] + dm["r"]) * p + 8 * (dm["n"] * dm["d"] ** 2 + dm["r"])
        # the certificate operator (Q - Lambda) X reads Q, X and the Lambda blocks and writes the result: it never reads
        # the point's rows, so it is NOT charged b_hvp (round 4 did, and printed 94-103 % of the peak)
        b_cert = b_spmm + 8 * (dm["n"] * dm["d"] ** 2 + dm["r"])
        row = "| %d | %d | %d |" % (n, p, c.ld)
        for fn, by in ((lambda: c.spmm_dev(x, p, o), b_spmm), (lambda: c.hvp_dev(x, o), b_hvp),
                       (lambda: c.certificate_product_dev(x, p, o), b_cert)):
            for _ in range(20): fn()
            c.sync(); c.timer_start()
            for _ in range(300): fn()
            us = c.timer_stop_ms() * 1e3 / 300
            frac = by / us / 1e3 / 80
            assert frac < 100.0, "a fraction of the roofline above 1 is a byte-count error"
            row += " %.2f | %.0f | %.1f%% |" % (us, by / us / 1e3, frac)
        print(row, flush=True)
        for q in (y, x, o): c.dev_free(q)

print("| poses | p | LD | Q·X µs | GB/s | of 8 TB/s | Hvp µs | GB/s | of 8 TB/s | (Q−Λ)X µs | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
if len(sys.argv) > 1:   # python tools/rank_sweep.py <poses> <rank> [<rank> ...]
    run(int(float(sys.argv[1])), [int(a) for a in sys.argv[2:]])
else:
    run(10000, [3, 4, 5, 6, 7, 10])
    run(100000, [3, 4, 5, 6, 7, 10, 12, 14, 16, 20])
