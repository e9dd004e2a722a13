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
typ = ((meta >> np.uint64(32)) & np.uint64(0xFF)).astype(np.int64)  # (a chain slice carries its tail sizes above)
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

# phases of the pose slices (stamps inside pose_slice): offsets from the wavefront's start
try:
    fp = ctx.L.cora_debug_spmm_phases
    fp.argtypes = [C.c_void_p, C.c_int]
    ph = np.zeros(6 * nb, dtype=np.uint64)
    assert fp(ph.ctypes.data, nb) == 0
    ph = ph.reshape(nb, 6)[ok].astype(np.int64)
    m = typ == 0
    names = ["windows landed", "fixed slots", "general slots", "tail", "projected"]
    for i, nm in enumerate(names):
        v = (ph[m, i] - t0[m]) / 100.0
        v = v[ph[m, i] > 0]
        if len(v):
            print("pose slices, %-16s after start: mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (nm, v.mean(), *np.percentile(v, [10, 50, 90, 100])))
    print("pose slices, end              after start: mean %.2f" % dur[m].mean())
    # where the pose wavefronts ran: HW_ID (wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) | XCC_ID << 32
    hw = ph[m, 5]
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (((hw >> 32) & 0xF) << 8)
    simd = (cu << 2) | ((hw >> 4) & 3)
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    per_simd = np.bincount(np.unique(simd, return_inverse=True)[1])
    print("pose wavefronts per CU: %d CUs used, histogram %s; per SIMD: %d SIMDs, histogram %s" % (
        len(per_cu), np.bincount(per_cu).tolist(), len(per_simd), np.bincount(per_simd).tolist()))
    endp = end[m]
    key = np.unique(cu, return_inverse=True)[1]
    by = [endp[key == k].max() for k in range(len(per_cu))]
    for cnt in sorted(set(per_cu.tolist())):
        sel = [b for b, c in zip(by, per_cu) if c == cnt]
        print("  CUs with %d pose wavefronts: %d, their last end mean %.2f max %.2f us" % (cnt, len(sel), np.mean(sel), np.max(sel)))
except AttributeError:
    pass
