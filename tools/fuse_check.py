"""Developer check: the three forms of the STPCG iteration (one explicit inverse | fused vector passes | one pass per operation)
from the same point of plaza2, by iteration count.  python tools/fuse_check.py [outer iterations before the comparison]"""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from cora_amd import capi, host
P = host.Problem.from_pyfg("/root/repo/tests/golden/datasets/plaza2.pyfg")
P.update(); P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY)
p = 4
P.set_rank(p); P.precond_info(); dm = P.dims()
h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
s, r, v, pk, hp, y = [h.dev_alloc(p) for _ in range(6)]
x0 = P.op("getRandomInitialGuess")
xs = P.tnt(x0, max_iterations=int(sys.argv[1]) if len(sys.argv) > 1 else 25)["x"]
h.upload(xs, y)
h.project_to_manifold_dev(y, y); h.set_point_dev(y)
grad = h.point_ptrs()[2]
res = {}
for mode, names in {"inverse": (), "fused": ("CORA_NO_INVERSE_FUSE",), "unfused": ("CORA_NO_FUSE",)}.items():
    for n_ in names: os.environ[n_] = "1"
    for it in (1, 2, 4, 8, 12, 16, 24):
        done, step = h.stpcg_dev(grad, 15.0, s, r, v, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=it)
        res[(mode, it)] = (done, step, h.download(s, p), h.stpcg_path())
    for n_ in names: os.environ.pop(n_, None)
for it in (1, 2, 4, 8, 12, 16, 24):
    a, b, c = res[("inverse", it)], res[("fused", it)], res[("unfused", it)]
    nb = np.abs(c[2]).max()
    print("its %2d paths %d %d %d done %d %d %d | step rel: inverse-unfused %.2e fused-unfused %.2e | s: inverse-unfused %.2e fused-unfused %.2e" % (
        it, a[3], b[3], c[3], a[0], b[0], c[0], abs(a[1] - c[1]) / abs(c[1]), abs(b[1] - c[1]) / abs(c[1]),
        np.abs(a[2] - c[2]).max() / nb, np.abs(b[2] - c[2]).max() / nb))
