"""Step 3 of fast_verification (ILDL-preconditioned LOBPCG) on partitions: iterations / theta with and without the
per-rank incomplete factor, against the single handle.  python tools/ildl_shard_probe.py [n]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cora_amd import capi, host
from cora_amd.dist import NativeLocalComm, NativeLocalGroup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
p = 4


def make():
    P, gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=4, n_ranges=n, seed=23, precond=capi.PRECOND_JACOBI, ground_truth=True)
    P.update()
    P.set_rank(p)
    return P, gt


P1, gt = make()
pert = float(os.environ.get("PERT", "0.05"))
Y0 = P1.op("projectToManifold", np.hstack([gt, np.zeros((gt.shape[0], p - gt.shape[1]))]) + pert * np.random.default_rng(1).standard_normal((gt.shape[0], p)))
iters = int(os.environ.get("TNT_ITERS", "3"))
Y = P1.tnt(Y0, max_iterations=iters)["x"] if iters > 0 else Y0
for eta in [float(x) for x in os.environ.get("ETAS", "1e-2,1e-4").split(",")]:
    for seed, ildl in ((True, True), (False, True), (False, False)):
        P1.set_verification_lab(seed=seed, ildl=ildl)
        c = P1.certify(Y, eta, nx=6)
        print("single  eta=%g seed=%d ildl=%d: certified=%s theta=%.4e iters=%d step3=%s" % (eta, seed, ildl, c["is_certified"], c["theta"], c["iters"], P1.certification_reached_step3()), flush=True)
    for world in (2, 4):
        group = NativeLocalGroup(world)
        out = [None] * world
        err = []

        def run(r):
            try:
                P, _ = make()
                P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
                res = []
                for seed, ildl in ((False, True), (False, False)):
                    P.set_verification_lab(seed=seed, ildl=ildl)
                    c = P.certify(Y, eta, nx=6)
                    res.append((seed, ildl, c["is_certified"], c["theta"], c["iters"], P.certification_reached_step3()))
                out[r] = res
            except BaseException as e:
                err.append(e)
                group.barrier.abort()
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join(600) for t in th]
        if err:
            raise err[0]
        for res in out[:1]:
            for x in res:
                print("world %d eta=%g seed=%d ildl=%d: certified=%s theta=%.4e iters=%d step3=%s" % ((world, eta) + x), flush=True)
