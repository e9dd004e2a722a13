"""Wavefront timeline of one Hvp launch (measurement build: CORA_EXTRA_HIPCC_FLAGS=-DCORA_SPMM_TIMES python cora_amd/build.py --force).
usage: python tools/spmm_timeline.py [poses] [rank]   -- prints when wavefronts start, how long they live, by slice type."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cora_amd import capi, host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P, _ = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, ground_truth=True)
P.update()
dm = P.dims()
_, _, rowptr, colidx, vals = P.matrix("DataMatrix")
ctx = capi.Context(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], rowptr, colidx, vals, device=0)
ctx.set_rank(p)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(7)
dev = torch.device("cuda", 0)
y = torch.zeros(ctx.rows * ctx.ld, dtype=torch.float64, device=dev)
x = torch.zeros_like(y)
out = torch.zeros_like(y)
ctx.upload(rng.uniform(-1, 1, (dm["N"], p)), y.data_ptr())
ctx.project_to_manifold_dev(y.data_ptr(), y.data_ptr())
ctx.set_point_dev(y.data_ptr())
ctx.upload(rng.uniform(-1, 1, (dm["N"], p)), x.data_ptr())
ctx.tangent_space_projection_dev(x.data_ptr(), x.data_ptr())
for _ in range(50):
    ctx.hvp_dev(x.data_ptr(), out.data_ptr())
torch.cuda.synchronize()
st = ctx.format_stats()
nb = ((st["long_chunks"] + 7) // 8) * 8 + ((st["slices"] + 7) // 8) * 8
buf = np.zeros(3 * nb, dtype=np.uint64)
fn = ctx.L.cora_debug_spmm_times
fn.argtypes = [C.c_void_p, C.c_int]
rc = fn(buf.ctypes.data, nb)
assert rc == 0, rc
t = buf.reshape(nb, 3)
ok = t[:, 1] > 0
t0 = t[ok, 0].astype(np.int64)
t1 = t[ok, 1].astype(np.int64)
meta = t[ok, 2]
typ = (meta >> np.uint64(32)).astype(np.int64)
wid = (meta & np.uint64(0xFFFFFFFF)).astype(np.int64)
base = t0.min()
start = (t0 - base) / 100.0   # us (100 MHz clock)
end = (t1 - base) / 100.0
dur = end - start
print("blocks %d (recorded %d), kernel span %.2f us" % (nb, ok.sum(), end.max()))
print("start   percentiles 10/50/90/99/max: %s" % np.round(np.percentile(start, [10, 50, 90, 99, 100]), 2))
print("end     percentiles 10/50/90/99/max: %s" % np.round(np.percentile(end, [10, 50, 90, 99, 100]), 2))
for ty in sorted(set(typ.tolist())):
    m = typ == ty
    print("type %3d: %5d waves, width mean %.1f max %d, duration mean %.2f p50 %.2f p90 %.2f max %.2f us, start mean %.2f max %.2f, end max %.2f"
          % (ty, m.sum(), wid[m].mean(), wid[m].max(), dur[m].mean(), np.median(dur[m]), np.percentile(dur[m], 90), dur[m].max(),
             start[m].mean(), start[m].max(), end[m].max()))
# resident waves over time
grid = np.arange(0, end.max(), 1.0)
res = [(int(((start <= g) & (end > g)).sum())) for g in grid]
print("resident waves at each us:", res)
# by block index: start time of every 256th block
idx = np.nonzero(ok)[0]
print("start by block index (every 256th):", [(int(i), round(float(s), 2)) for i, s in zip(idx[::256], start[::256])])
