"""Predicted microseconds per phase of a partitioned Hessian-vector product and of a sharded STPCG iteration at N = 2 / 4 / 8
GPUs, from the ACTUAL partitions of the bench graphs (built on the CPU: format builder only, no device) and the figures of
/opt/skills/guides/MI355X_MICROARCH.md + the task statement's link rate.  Writes the table the first SCALE record is to be
read against (profiles/r06_multi_gpu_model.md).      python tools/multi_gpu_model.py [poses ...]

Round 6: two transports -- the RCCL one (an ASSUMED latency band per small collective) and the device-side one (cora_comm_create_p2p:
a collective is one small kernel that pushes into the peers' mailboxes over xGMI, sets per-peer flags and spins on its own; the
hand-off is priced from the guide's 1-to-1 flag hand-off plus one xGMI store)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cora_amd import capi, host

# --- figures ------------------------------------------------------------------------------------------------------------
LINK_GBS = 153.0          # one xGMI link, one direction (task statement: 7 links x ~153 GB/s per GPU)
BOUNDARY_US = 1.7         # dependent kernel boundary between real kernels (guide, price table row "boundary": 1.45-1.9)
SMALL_KERNEL_US = 3.0     # a pack / unpack / scalar-step launch of a few hundred blocks: boundary + one dependent round trip or two
RCCL_LAT_US = (12.0, 25.0)  # ASSUMPTION (not in the guides): latency of one small ncclAllGather / ncclAllReduce on a stream at
                            # N = 2 .. 8 (ring / one-shot protocols on xGMI, LL128): the number the first SCALE record pins
P2P_HANDOFF_US = (2.0, 4.0)  # ASSUMPTION: what the push + flag + the peers' flags cost one exchange / all-reduce kernel ON TOP of a small
                             # launch (guide, row "handoff": 1-3 us between blocks of one GPU; + one xGMI store and its flag, ~1 us)
SINGLE = {100000: dict(hvp_us=17.1, hvp_loop_us=18.6, stpcg_us=110.8, sweeps_top_us=90.3),    # measured, one MI355X (r06 bench)
          1000000: dict(hvp_us=170.5, hvp_loop_us=177.0, stpcg_us=1110.0, sweeps_top_us=930.0)}   # (r05 bench at 10^6 poses)
P = 5


def partition_stats(n, world):
    Pr = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42)
    Pr.update()
    dm = Pr.dims()
    _, _, rowptr, colidx, vals = Pr.matrix("DataMatrix")
    need, slices, nnz, shard = [], [], [], 0
    for r in range(world):
        c = capi.Context(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], rowptr, colidx, vals, device=-1, rank=r, world=world)
        need.append(np.asarray(c.remote_rows()))
        st = c.format_stats()
        slices.append(st["slices"] if "slices" in st else 0)
        nnz.append(st["local_nnz"])
        shard = c.shard_rows
        n_long = len(c.long_rows())
        m = c.row_map()
    # exports of rank q: rows of its shard some other rank reads; the exchange is padded to the longest list
    exports = [set() for _ in range(world)]
    for r in range(world):
        for row in need[r]:
            exports[int(row) // shard].add(int(row))   # (internal rows: rank-major shards)
    e_max = max(len(e) for e in exports)
    return dict(N=dm["N"], shard_rows=shard, e_max=e_max, n_long=n_long, nnz_max=max(nnz), nnz=sum(nnz), slices_max=max(slices))


def model(n, world, st, transport="rccl"):
    s = SINGLE[n]
    ld = P
    msg = (st["e_max"] + st["n_long"]) * ld * 8          # bytes each rank contributes to the product's all-gather
    wire = msg * (world - 1) / (LINK_GBS * 1e3)           # us on the wire if the ranks' pieces arrive over one link each, in turn
    frac = st["nnz_max"] / st["nnz"]
    kern = max(s["hvp_loop_us"] * frac, 4.0)              # the slices of the busiest rank (no product of this kernel runs below ~4 us)
    chunks = 3.0                                          # the long rows' chunks on this rank's columns: 10 rows, a launch of its own
    if transport == "p2p":
        # chunks | ONE exchange kernel (exported rows read from X, push, hand-over, wait, unpack from the mailbox) | slices: no
        # pack launch;  a reduction is one
        # kernel too (push, hand-over, wait, add in rank order) followed by the scalar step's launch
        lo = chunks + SMALL_KERNEL_US + P2P_HANDOFF_US[0] + wire + kern
        hi = chunks + SMALL_KERNEL_US + P2P_HANDOFF_US[1] + wire + kern
        sweeps = max(s["sweeps_top_us"] * frac, 55.0)
        red_lo, red_hi = SMALL_KERNEL_US + P2P_HANDOFF_US[0], SMALL_KERNEL_US + P2P_HANDOFF_US[1]
        it_lo = lo + red_lo + SMALL_KERNEL_US + sweeps + red_lo + SMALL_KERNEL_US
        it_hi = hi + red_hi + SMALL_KERNEL_US + sweeps + red_hi + SMALL_KERNEL_US
        return dict(msg=msg, wire=wire, kern=kern, lo=lo, hi=hi, it_lo=it_lo, it_hi=it_hi, sweeps=sweeps)
    lo = SMALL_KERNEL_US + chunks + RCCL_LAT_US[0] + wire + SMALL_KERNEL_US + kern
    hi = SMALL_KERNEL_US + chunks + RCCL_LAT_US[1] + wire + SMALL_KERNEL_US + kern
    # sharded STPCG iteration (block Jacobi over the ranks, sweep-fused per shard): product + kappa all-reduce + scalar step +
    # the shard's sweeps and last stage (1 / N of the single factor's work, not below the chain of one block ~ 25 us) +
    # one all-reduce of two doubles + scalar step
    sweeps = max(s["sweeps_top_us"] * frac, 55.0)
    it_lo = lo + RCCL_LAT_US[0] + SMALL_KERNEL_US + sweeps + RCCL_LAT_US[0] + SMALL_KERNEL_US
    it_hi = hi + RCCL_LAT_US[1] + SMALL_KERNEL_US + sweeps + RCCL_LAT_US[1] + SMALL_KERNEL_US
    return dict(msg=msg, wire=wire, kern=kern, lo=lo, hi=hi, it_lo=it_lo, it_hi=it_hi, sweeps=sweeps)


def main():
    sizes = [int(float(a)) for a in sys.argv[1:]] or [100000, 1000000]
    print("# Round 6 -- what the first multi-GPU run is expected to show (model, nothing here has run on more than one GPU)\n")
    print("`python tools/multi_gpu_model.py`: the partitions are the real ones (format builder on the CPU, `cora_ctx_create_part` with "
          "`device = -1`), the times are a model from: one xGMI link %.0f GB/s per direction (task statement), dependent kernel "
          "boundary %.1f us and ~%.0f us for a small pack / unpack / scalar-step launch (guide, price table rows *boundary*, "
          "*handoff*), the single-GPU kernels of this round's bench (`SINGLE` in the script), and -- the one number neither guide "
          "holds -- **%.0f-%.0f us assumed for one small RCCL collective on a stream**.  `bench.py --gpus N` prints the same "
          "phases measured (`multi_gpu.phases_us`: pack | long-row chunks | all-gather | unpack | slices), the collectives per "
          "product and per STPCG iteration, and `rccl_ranks` = ncclCommCount.\n" % (LINK_GBS, BOUNDARY_US, SMALL_KERNEL_US, *RCCL_LAT_US))
    for n in sizes:
        print("## %d poses, p = %d (single GPU: Hvp %.1f us back to back, %.1f us in the loop; STPCG iteration %.0f us)\n" % (
            n, P, SINGLE[n]["hvp_us"], SINGLE[n]["hvp_loop_us"], SINGLE[n]["stpcg_us"]))
        print("| N | rows per shard | rows exchanged per rank (longest export list) + slots | bytes per rank | pack | chunks | all-gather (latency + wire) | unpack | slices of the busiest rank | **product** | Hvp/s (whole job) | speed-up vs 1 GPU (in loop) | STPCG iteration (1 all-gather + 2 all-reduces) |")
        print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
        stats = {}
        for world in (2, 4, 8):
            st = stats[world] = partition_stats(n, world)
            m = model(n, world, st)
            print("| %d | %d | %d + %d | %d | %.0f | %.0f | %.0f-%.0f + %.2f | %.0f | %.1f | **%.0f-%.0f us** | %.0f-%.0f k | %.2f-%.2f x | %.0f-%.0f us (sweeps + last stage of a shard: %.0f) |" % (
                world, st["shard_rows"], st["e_max"], st["n_long"], m["msg"], SMALL_KERNEL_US, 3.0, RCCL_LAT_US[0], RCCL_LAT_US[1], m["wire"],
                SMALL_KERNEL_US, m["kern"], m["lo"], m["hi"], 1e3 / m["hi"], 1e3 / m["lo"], SINGLE[n]["hvp_loop_us"] / m["hi"],
                SINGLE[n]["hvp_loop_us"] / m["lo"], m["it_lo"], m["it_hi"], m["sweeps"]), flush=True)
        print()
        print("The same with the device-side transport (`cora_comm_create_p2p`: no pack launch, the exchange is ONE kernel, priced at a small launch + "
              "%.0f-%.0f us of hand-off; a reduction likewise):\n" % P2P_HANDOFF_US)
        print("| N | chunks | exchange kernel (launch + hand-off + wire) | slices of the busiest rank | **product** | Hvp/s (whole job) | speed-up vs 1 GPU (in loop) | STPCG iteration |")
        print("|---|---|---|---|---|---|---|---|")
        for world in (2, 4, 8):
            m = model(n, world, stats[world], "p2p")
            print("| %d | %.0f | %.0f + %.0f-%.0f + %.2f | %.1f | **%.0f-%.0f us** | %.0f-%.0f k | %.2f-%.2f x | %.0f-%.0f us |" % (
                world, 3.0, SMALL_KERNEL_US, P2P_HANDOFF_US[0], P2P_HANDOFF_US[1], m["wire"], m["kern"], m["lo"], m["hi"],
                1e3 / m["hi"], 1e3 / m["lo"], SINGLE[n]["hvp_loop_us"] / m["hi"], SINGLE[n]["hvp_loop_us"] / m["lo"], m["it_lo"], m["it_hi"]), flush=True)
        print()
    print("Reading it: the exchange is the chain halo plus the landmark slots -- a few hundred bytes to a few kilobytes per rank --, "
          "so the wire time is nil and a product costs its **launches and the collective's latency**: at 10^5 poses the model puts every "
          "N below the single GPU (speed-up < 1, as SURVEY 8e predicted), at 10^6 poses N = 8 is expected to come out ahead once "
          "the busiest rank's slices (1/8 of 258 us) outweigh ~40 us of fixed cost.  If the measured all-gather phase is far above "
          "the assumed band, the device-side exchange (second table; built in round 6, `bench.py --gpus N` times both and quotes the "
          "faster) is what takes its place; if `slices_us` is far above 1/N of the single-GPU kernel, the partition's nnz balance (printed by the bench "
          "as `local_nnz`) is the first suspect.  The interior / boundary overlap of the exchange (2 048 interior slices per rank "
          "on: 10^6 poses at N <= 8... 4 882 per rank at N = 8) hides the all-gather behind the interior slices and is what the 10^6-pose "
          "column should show as a product close to `slices + pack + chunks`.")


if __name__ == "__main__":
    main()
