"""Overlap of the exchange with the interior slices on partitioned handles (one GPU, ranks = threads of this process,
the library's own communication with the in-process transport).  Every rank runs `reps` Hessian-vector products on
resident vectors, once in the overlapped form and once in the serial one; prints the time per product.  Under
`rocprofv3 --kernel-trace` the trace shows, per launching thread (= rank), k_move_rows (pack / scatter) inside the time
span of that rank's interior k_spmm launch.

python tools/overlap_probe.py [poses] [world] [reps]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host
from cora_amd.dist import NativeLocalComm, NativeLocalGroup

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
p = 5
group = NativeLocalGroup(world)
rng = np.random.default_rng(3)
P0 = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_JACOBI)
P0.update(); P0.set_rank(p)
dm = P0.dims()
Yh = rng.uniform(-1, 1, (dm["N"], p)); Vh = rng.uniform(-1, 1, (dm["N"], p))
out = [None] * world
meet = threading.Barrier(world)


def body(r):
    P = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, precond=capi.PRECOND_JACOBI)
    P.update(); P.set_rank(p)
    comm = P.set_partition(r, world, lambda ctx: NativeLocalComm(ctx, group))
    ctx = comm.ctx
    y, x, o = ctx.dev_alloc(p), ctx.dev_alloc(p), ctx.dev_alloc(p)
    ctx.upload(Yh, y); ctx.project_to_manifold_dev(y, y); ctx.set_point_dev(y)
    ctx.upload(Vh, x); ctx.tangent_space_projection_dev(x, x)
    res = {}
    for mode in ((True,) if os.environ.get("CORA_PROBE_ONLY_SPLIT") else (True, False, True, False)):
        comm.overlap(2 if mode else 0)
        for _ in range(10):
            ctx.hvp_dev(x, o)
        ctx.sync(); meet.wait()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.hvp_dev(x, o)
        ctx.sync(); meet.wait()
        res.setdefault(mode, []).append((time.perf_counter() - t0) / reps * 1e6)
    out[r] = (res, comm.exchanged_rows, ctx.shard_rows)


th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
for t in th: t.start()
for t in th: t.join()
res, ex, shard = out[0]
print("%d poses, %d partitions on one GPU, %d exchanged rows of %d per shard" % (n, world, ex, shard))
print("Hvp per step, all ranks in step: overlapped %s us | serial %s us" % (
    ", ".join("%.1f" % v for v in res[True]), ", ".join("%.1f" % v for v in res.get(False, []))))
