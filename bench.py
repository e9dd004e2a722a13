#!/usr/bin/env python3
"""Headline benchmark: Riemannian Hessian-vector products per second on the
synthetic 10^5-pose RA-SLAM graph at relaxation rank p = 5 (BASELINE.json).

A "step" is one Hvp  out = Proj_Y((Q - Lambda) Ydot)  (reference
src/CORA_problem.cpp:822-867) with every operand resident in HBM:
  * N = 1: one launch of the fused SpMM+epilogue kernel;
  * N > 1: the data matrix is row-partitioned over the ranks (pose-aligned,
    nnz-balanced); each step is an RCCL all-gather of the search direction over
    xGMI followed by the local kernel (strong scaling on the fixed graph).

Prints ONE JSON line on rank 0.  Launch for N > 1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(d, n, r, N, nnz, p):
    """SURVEY 8(d) / BASELINE.md section 3 (CSR int32/fp64, X read once, result written once)."""
    b_spmm = 12 * nnz + 4 * (N + 1) + 16 * N * p
    b_hvp = b_spmm + 8 * (d * n + r) * p + 8 * (n * d * d + r)
    return b_spmm, b_hvp


def pmc_traffic(args, ld, epi):
    """HBM-side bytes per launch of the timed kernels, collected now: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE --
    separate passes, no trace domains, as MI355X_MICROARCH.md prescribes) of a short child run of this command that
    launches the back-to-back products and then the STPCG loop and stops.  gfx950: FETCH_SIZE counts 128-byte fabric
    requests at 64 bytes, so reads = 2 x FETCH_SIZE (calibrated with a streaming kernel, profiles/hvp_traffic.json); both
    counters are in KB.  Returns {kernel name: {"read", "write", "launches"}} (bytes per launch, averaged over the
    launches of that kernel) or None when rocprofv3 is missing, this run is itself under a profiler, or a pass fails."""
    import csv, glob, shutil, subprocess, tempfile
    if os.environ.get("CORA_BENCH_CHILD") or shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    regex = "k_spmm<%d, 3, [0-3]>|k_subblock<%d|k_rowop<%d|k_kappa_finish" % (ld, ld, ld)
    env = dict(os.environ, CORA_BENCH_CHILD="1", TMPDIR="/tmp")
    per = {}
    tmp = tempfile.mkdtemp(prefix="cora_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-include-regex", regex, "--output-format", "csv", "-d", out,
                   "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--kernel-only", "--steps", "60",
                   "--warmup", "5", "--poses", str(args.poses), "--rank", str(args.rank), "--op", args.op,
                   "--pmc-traffic", "off"]
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("cora::", "").strip()
                t = acc.setdefault(name, [0.0, 0])
                t[0] += float(row["Counter_Value"])
                t[1] += 1
            for name, (tot, cnt) in acc.items():
                e = per.setdefault(name, {"launches": cnt})
                kb = tot / cnt
                e["read" if counter == "FETCH_SIZE" else "write"] = int(round((2.0 if counter == "FETCH_SIZE" else 1.0) * kb * 1024))
    except Exception:  # noqa: BLE001 -- a failed counter pass must not fail the bench: the committed figure is reported instead
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per = {k: v for k, v in per.items() if "read" in v and "write" in v}
    return per or None


def rocprof_kernel_times(args, ld):
    """Average duration of every kernel of the loop as rocprofv3 sees it: one `--kernel-trace --stats` pass of the same short
    child run the counter passes use.  HIP events around single launches INSIDE the solver loop carry their own cost on the
    stream (2-5 us each, more than a kernel boundary), so the in-loop figures of the line come from here; the event-based
    ones stay beside them.  Returns {kernel name: {"us", "calls"}} or None (no rocprofv3 / already under a profiler)."""
    import csv, glob, shutil, subprocess, tempfile
    if os.environ.get("CORA_BENCH_CHILD") or shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    env = dict(os.environ, CORA_BENCH_CHILD="1", TMPDIR="/tmp")
    tmp = tempfile.mkdtemp(prefix="cora_kt_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "kt", "--", sys.executable,
               os.path.abspath(__file__), "--kernel-only", "--steps", "60", "--warmup", "5", "--poses", str(args.poses),
               "--rank", str(args.rank), "--op", args.op, "--pmc-traffic", "off"]
        r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        out = {}
        for row in csv.DictReader(open(files[0])):
            name = row["Name"].split("(")[0].replace("void ", "").replace("cora::", "").strip()
            out[name] = {"us": float(row["AverageNs"]) / 1e3, "calls": int(row["Calls"])}
        return out or None
    except Exception:  # noqa: BLE001 -- a failed profiler pass must not fail the bench
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


PMC_HOW = ("collected by this run: two rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; separate passes, no trace "
           "domains) of a short child run of this command, per-launch averages over the launches of each kernel")


def cpu_quota():
    """CPUs the container may use: cgroup v2 cpu.max / v1 cfs quota, else the CPUs it may run on."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return len(os.sched_getaffinity(0))


def cpu_baseline(rowptr, colidx, vals, dm, p, budget_s, threads=1):
    """The oracle's Hvp (kind "port") on the same workload.  threads=1 is the reference-equivalent
    baseline: the reference itself is single-threaded (no OpenMP in its CMakeLists.txt)."""
    from oracle import oracle as orc
    orc.set_threads(threads)
    Q = orc.CSR(rowptr, colidx, vals, dm["N"])
    dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
    rng = np.random.default_rng(7)
    Y = orc.project_manifold(dims, rng.uniform(-1, 1, (dm["N"], p)))
    G = orc.egrad(Q, Y)
    V = orc.tangent_proj(dims, Y, rng.uniform(-1, 1, (dm["N"], p)))
    out = work = None
    if threads > 1:
        # threads spread over the sockets and bound to their CPUs; every thread's rows of Q and of the dense operands on
        # its own memory node (first touch with the loops' own schedule)
        orc.bind_threads(threads)
        Q = orc.numa_csr(Q)
        Y, G, V = orc.numa_dense(Y), orc.numa_dense(G), orc.numa_dense(V)
        out, work = orc.numa_dense(shape=Y.shape), orc.numa_dense(shape=Y.shape)
    ref = orc.hvp(Q, dims, Y, G, V, out=out, work=work)
    if out is not None:
        ref = ref.copy()
    t0 = time.perf_counter()
    orc.hvp(Q, dims, Y, G, V, out=out, work=work)  # (the first call pays for cold caches and, threaded, the team's start)
    one = time.perf_counter() - t0
    reps = max(3, min(2000, int(budget_s / max(one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.hvp(Q, dims, Y, G, V, out=out, work=work)
    dt = time.perf_counter() - t0
    if threads > 1:
        orc.bind_threads(0)
    return reps / dt, reps, (Y, V, ref)


def stpcg_setup(P, dm, p, x_gt):
    """The C++ host's own handle with the reference's default preconditioner (RegularizedCholesky) installed and the
    generator's ground truth (padded to rank p) as the current point: there the Hessian is positive semidefinite up to
    the noise, so the truncated CG is not cut short by negative curvature.  Returns (handle, six resident vectors)."""
    from cora_amd import capi
    P.set_rank(p)
    t0 = time.perf_counter()
    P.context_ptr()   # the Problem's own device handle (format of Q, uploads): not part of the preconditioner's set-up
    t_handle = time.perf_counter() - t0
    t0 = time.perf_counter()
    P.set_preconditioner(capi.PRECOND_REGULARIZED_CHOLESKY)
    info = P.precond_info()
    t_setup = time.perf_counter() - t0
    h = capi.Context.from_handle(P.context_ptr(), dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"])
    v = [h.dev_alloc(p) for _ in range(7)]
    Yh = np.zeros((dm["N"], p))
    Yh[:, :dm["d"]] = x_gt
    h.upload(Yh, v[5])
    h.project_to_manifold_dev(v[5], v[5])
    h.set_point_dev(v[5])
    # what TNT holds when it starts an inner solve (the accepted point's wait returns them, cora_tnt_accept_dev): P g and
    # the inner products <g, g>, <g, P g> -- the solve then starts without a preconditioner apply of its own
    grad = h.point_ptrs()[2]
    h.precondition_projected_dev(grad, v[6])
    h.warm = h.dots_dev([(grad, grad), (grad, v[6])])
    return h, v, info, t_handle, t_setup


def stpcg_run(h, v, its):
    """`its` iterations of the solver's own inner loop: cora_stpcg_dev from the Riemannian gradient at the current
    point, tolerance out of reach and a huge radius so that it runs exactly `its` iterations (Hvp + update +
    preconditioner + reductions + direction, scalars on the device).  Returns (iterations done, seconds)."""
    s, r, z, pk, hp, _, pg = v
    grad = h.point_ptrs()[2]
    h.sync()
    t0 = time.perf_counter()
    # (cora_stpcg_warm_dev: the entry TNT calls -- host/TNT.cpp -- with P g and its inner products from the accepted point)
    done, _ = h.stpcg_warm_dev(grad, pg, h.warm[0], h.warm[1], 1e30, s, r, z, pk, hp, kappa_fgr=1e-300, theta=0.0, max_iters=its)
    h.sync()
    return done, time.perf_counter() - t0


def stpcg_bytes(dm, p, nnz_L, ent):
    """Algorithmic bytes of one sweep-fused STPCG iteration (same convention as B_spmm: 12 bytes per
    stored entry, every vector pass 8 N p): what each launch must move at least."""
    N, d, n, r = dm["N"], dm["d"], dm["n"], dm["r"]
    _, b_hvp = algorithmic_bytes(d, n, r, N, dm["nnz"], p)
    vec = 8 * N * p
    return {
        "product": b_hvp,
        "kappa": 0,
        # forward: L once; r and Hp in, r and y out
        "forward_sweep": 12 * nnz_L + 4 * vec,
        # the last stage's two products: their stored entries (explicit inverse + folded aux sums), rows negligible
        "top_forward": 12 * ent["top_forward"],
        "top_backward": 12 * ent["top_backward"],
        # backward: L once; y, p, s in, the point's rows (rotation + range rows) in, p and s out
        "backward_sweep": 12 * nnz_L + 5 * vec + 8 * (d * n + r) * p,
    }


def solver_loop_timings(P, ctx, dm, p, x, out, kernel_us, x_gt):
    """SURVEY 8(d) timing protocol beside the headline number: percentiles of single launches, the plain
    Q.X product, and one full STPCG iteration (Hvp + preconditioner + inner products + updates,
    src/CORA.cpp:71-92,119-122) with the reference's default RegularizedCholesky preconditioner."""
    b_spmm, _ = algorithmic_bytes(dm["d"], dm["n"], dm["r"], dm["N"], dm["nnz"], p)
    one = []
    for _ in range(200):
        ctx.timer_start()
        ctx.hvp_dev(x.data_ptr(), out.data_ptr())
        one.append(ctx.timer_stop_ms() * 1e3)
    one.sort()
    ctx.timer_start()
    for _ in range(300):
        ctx.spmm_dev(x.data_ptr(), p, out.data_ptr())
    spmm_us = ctx.timer_stop_ms() * 1e3 / 300
    ex = {
        "hvp_single_launch_us": {"p10": one[20], "p50": one[100], "p90": one[180],
                                 "note": "one launch between two events: includes the launch floor that back-to-back launches hide"},
        "spmm_us": spmm_us,
        "spmm_GBps": b_spmm / spmm_us / 1e3,
        "hvp_back_to_back_us": kernel_us,
    }
    # full STPCG iteration on the C++ host's own handle with the Cholesky preconditioner installed
    h, v, info, ex["problem_handle_s"], ex["preconditioner_setup_s"] = stpcg_setup(P, dm, p, x_gt)
    ex["preconditioner"] = {"kind": "RegularizedCholesky", "nnz_L": info["nnz"], "lambda": info["lam"]}
    s, r, z, pk, hp, y, _pg = v
    its = 60
    stpcg_run(h, v, 8)
    done, dt = stpcg_run(h, v, its)
    ex["stpcg_iteration_us"] = dt / max(done, 1) * 1e6
    ex["stpcg_iterations_timed"] = done
    done_l, dt_l = stpcg_run(h, v, 240)   # the same with four times the iterations: what a solve's start and end weigh
    ex["stpcg_iteration_us_240"] = dt_l / max(done_l, 1) * 1e6
    ex["stpcg_form"] = {0: "one pass per operation", 1: "fused vector passes (5 launches + the solve)",
                        3: "one explicit inverse: product | kappa + residual | W | W^T | projection + step + direction",
                        2: "sweep-fused: product with the kappa partials | [kappa: a launch of its own above 4 096 partials, else "
                           "added by every block of the forward sweep] | forward sweep (r += alpha Hp, <r,r> and |L^-1 r|^2 "
                           "slots) | last stage, two products (the second finishes <r,r> and <r,v> = |L^-1 r|^2) | backward "
                           "sweep (v = Proj_Y(x), s += alpha p, p = -v + beta p)"}.get(h.stpcg_path(), "?")
    # the launches of the iteration one by one (HIP events around every launch; the events cost the stream a little, so
    # the sum is reported beside the clean iteration above, not instead of it).  The product's figure is the Hessian-
    # vector product as it runs INSIDE the loop: two sweeps over the factor (130 MB, non-temporal loads) pass between
    # two products
    h.profile_stpcg(2)
    stpcg_run(h, v, its)
    ex["stpcg_phase_us"] = h.stpcg_phase_us()
    h.profile_stpcg(1)   # events around the product only (+ two in a row at the iteration's end: what an event costs)
    stpcg_run(h, v, its)
    raw, ex["hvp_in_stpcg_samples"] = h.stpcg_hvp_us()
    ov = h.stpcg_phase_us().get("event_overhead") or 0.0
    # the interval between two events holds the kernel AND one event's own cost on the stream (2-4 us: two events with
    # nothing between them measure it in the same run); the kernel alone is what rocprofv3 reports for k_spmm<LD, 3, 3>
    ex["hvp_in_stpcg_us"] = max(raw - ov, 0.0)
    ex["hvp_in_stpcg_with_event_us"] = raw
    ex["event_overhead_us"] = ov
    h.profile_stpcg(0)
    ex["_stpcg_entries"] = h.precond_entries()
    h.timer_start()
    for _ in range(50):
        h.precondition_projected_dev(r, z)
    ex["preconditioner_apply_us"] = h.timer_stop_ms() * 1e3 / 50
    for q in v:
        h.dev_free(q)
    # end to end at this size: the full staircase (solveCORA, rank 3 upwards, certification by factorisation) from
    # the generator's ground truth -- where a front end would leave the problem; what the host does per level
    # (numeric LL^T + solve plan of the preconditioner, LL^T of the certificate) is inside these seconds
    t0 = time.perf_counter()
    P.set_rank(dm["d"])
    res = P.solve(P.op("projectToManifold", x_gt), max_rank=7, max_seconds=120)
    ex["staircase_from_ground_truth"] = {
        "seconds": time.perf_counter() - t0, "solver_seconds": res["seconds"], "hessian_vector_products": res["hvps"],
        "levels": res["levels"], "f": res["f"], "grad_norm": res["grad_norm"], "certified": bool(res["certified"]),
        "theta": res["theta"], "eta": res["eta"], "chi_square_sized_optimum": dm["r"] / 2}
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--poses", type=int, default=100000)
    ap.add_argument("--rank", type=int, default=5, help="relaxation rank p")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--op", choices=["hvp", "cert"], default="hvp",
                    help="hvp: Riemannian Hessian-vector product (the headline metric); cert: certificate operator "
                         "(Q - Lambda) X with --rank columns (BASELINE config 5, use --rank 10)")
    ap.add_argument("--pmc-traffic", choices=["auto", "off"], default="auto",
                    help="auto: HBM-side bytes of the timed kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) "
                         "of a short child run of this command, collected by this run (N = 1; skipped under a profiler)")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="the all-core column with a sweep over several thread counts (~10 s of host time); by default only "
                         "two thread counts are tried (the container's CPU quota and half of it, ~1 s each)")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-core column altogether")
    ap.add_argument("--kernel-only", action="store_true", help=argparse.SUPPRESS)  # the child run of --pmc-traffic
    args = ap.parse_args()

    # stdout carries ONE line, the JSON result: whatever the libraries print on the way (the C++ host reports like the
    # reference does, on std::cout) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from cora_amd import capi, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the CORA hot path has no CPU fallback")
    # CORA_BENCH_BACKEND=gloo lets several ranks share one GPU (functional check of the N > 1 path
    # on a 1-GPU box); the driver's runs use the default "nccl" (= RCCL) with one GPU per rank.
    backend = os.environ.get("CORA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    # ---- workload: the C++ host generates the graph and assembles Q ----------
    n, p = args.poses, args.rank
    P, x_gt = host.Problem.synthetic(dim=3, n_poses=n, n_landmarks=10, n_ranges=n // 2, seed=42, ground_truth=True)
    P.update()
    dm = P.dims()
    _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
    ctx = capi.Context(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], rowptr, colidx, vals,
                       device=local_rank, rank=rank, world=world)
    ctx.set_rank(p)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ld, rows, shard = ctx.ld, ctx.rows, ctx.shard_rows
    b_spmm, b_hvp = algorithmic_bytes(dm["d"], dm["n"], dm["r"], dm["N"], dm["nnz"], p)

    # operands: U(-1,1) seed 7, Y on the manifold, Ydot in T_Y
    rng = np.random.default_rng(7)
    Yh = rng.uniform(-1, 1, (dm["N"], p))
    Vh = rng.uniform(-1, 1, (dm["N"], p))
    dev = torch.device("cuda", local_rank)
    y = torch.zeros(rows * ld, dtype=torch.float64, device=dev)
    x = torch.zeros(rows * ld, dtype=torch.float64, device=dev)
    out = torch.zeros(rows * ld, dtype=torch.float64, device=dev)
    # N > 1: the handle is collective -- cora_amd.dist.TorchComm installs the three injected steps (cora_set_comm):
    # before every product the rows of the operand that another rank's part of Q reads are packed
    # (cora_pack_rows_dev), moved by ONE RCCL all-gather and scattered (cora_scatter_rows_dev)
    comm = None
    transports = []   # N > 1: the transports this run measures, in order; the LAST one that works stays installed

    def agree(ok):    # every rank takes the same path
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev if backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return float(flag.item()) != 0.0

    def make_comm(kind):
        """p2p: device-side collectives over IPC-mapped mailboxes (cora_comm_create_p2p); rccl: the library's own RCCL
        communicator (torch only hands its 128-byte id round); torch: injected callbacks over torch.distributed."""
        from cora_amd.dist import NativeP2PComm, NativeRcclComm, TorchComm
        c_ = None
        try:
            if kind == "p2p":
                # (a peer that never arrives must cost seconds, not the lease: the waiting kernels give up after this long,
                # and a transport that has raised a timeout while it was set up is dropped for the next one)
                os.environ.setdefault("CORA_P2P_TIMEOUT_S", "5")
                c_ = NativeP2PComm(ctx)
                # pre-flight, before anything else depends on it: three products (exchange of the operand) and an inner
                # product (all-reduce) on scratch vectors; a transport that times out or fails here is dropped for the next
                # one while every rank is still at the same point of the script (no torch collective in between)
                a_, b_ = ctx.dev_alloc(p), ctx.dev_alloc(p)
                try:
                    for _ in range(3):
                        ctx.spmm_dev(a_, p, b_)
                    ctx.dot_dev(a_, b_, p)
                    ctx.sync()
                finally:
                    ctx.dev_free(a_)
                    ctx.dev_free(b_)
                st_ = c_.status()
                if st_["timeouts"] != 0:
                    raise RuntimeError("a waiting kernel timed out during the set-up: %s" % st_)
            elif kind == "rccl":
                c_ = NativeRcclComm(ctx, device=dev)
                assert c_.nranks in (world, -1), "ncclCommCount = %d, WORLD_SIZE = %d" % (c_.nranks, world)
            else:
                c_ = TorchComm(ctx, device=dev)
        except Exception as e:  # noqa: BLE001 -- reported; the run goes on with the next transport
            sys.stderr.write("rank %d: %s transport failed (%s)\n" % (rank, kind, e))
            c_ = None
        if not agree(c_ is not None):
            return None
        return c_

    if world > 1:
        # CORA_BENCH_COMM: which transports to time (comma separated).  Default: the device-side one AND the RCCL one, so
        # that the first run on real links shows both; `value` is the faster of those whose product matches the CPU oracle.
        want = os.environ.get("CORA_BENCH_COMM", "p2p,rccl" if backend == "nccl" else "p2p,torch").split(",")
        transports = [k for k in want if k in ("p2p", "rccl", "torch")]
        if not transports:
            transports = ["torch"]
        comm = make_comm(transports[0])
        if comm is None:
            for k in (["rccl"] if backend == "nccl" else []) + ["torch"]:
                if k not in transports:
                    transports.append(k)
            transports.pop(0)
            while transports and comm is None:
                comm = make_comm(transports[0])
                if comm is None:
                    transports.pop(0)
        if comm is None:
            raise SystemExit("no transport could be created for %d ranks" % world)
    ctx.upload(Yh, y.data_ptr())
    ctx.project_to_manifold_dev(y.data_ptr(), y.data_ptr())   # row-local: every rank projects its own rows
    ctx.set_point_dev(y.data_ptr())                           # collective: exchanges Y, reduces the cost
    ctx.upload(Vh, x.data_ptr())
    ctx.tangent_space_projection_dev(x.data_ptr(), x.data_ptr())
    k_op = p
    if args.op == "cert":   # BASELINE config 5: the certificate operator (Q - Lambda) X, k columns
        xk = torch.zeros(rows * ctx.L.cora_ld_for(k_op), dtype=torch.float64, device=dev)
        ok = torch.zeros_like(xk)
        ctx.upload(rng.uniform(-1, 1, (dm["N"], k_op)), xk.data_ptr())

    def step():
        if args.op == "cert":
            ctx.certificate_product_dev(xk.data_ptr(), k_op, ok.data_ptr())
        else:
            ctx.hvp_dev(x.data_ptr(), out.data_ptr())

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_region():
        for _ in range(args.warmup):
            step()
        fence()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        fence()
        return time.perf_counter() - t0_

    elapsed = timed_region()
    transport_runs = []
    if world > 1:
        # every transport in turn on the same handle (a new communicator replaces the old one); the product of each is
        # checked against the first one's -- all of them add in rank order, so the bits must be the same
        def product_now():
            ctx.hvp_dev(x.data_ptr(), out.data_ptr()) if args.op == "hvp" else step()
            torch.cuda.synchronize()
            return (out if args.op == "hvp" else ok).clone()
        first = product_now()

        def record(kind, c_, el):
            t_ = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else torch.device("cpu"))
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            now = product_now()
            rel = float((now - first).abs().max() / first.abs().max())
            # (the library's transports add the long rows' slots in rank order: the same bits; injected callbacks leave the
            # order of that sum to torch.distributed's all-reduce: rounding)
            e = {"transport": kind, "class": type(c_).__name__, "step_us": float(t_.item()) / args.steps * 1e6,
                 "same_bits_as_first": bool(torch.equal(now, first)), "max_rel_diff_to_first": rel}
            e["ok"] = rel < 1e-12
            if hasattr(c_, "status"):
                e["p2p_status"] = c_.status()
                e["ok"] = e["ok"] and e["p2p_status"]["timeouts"] == 0
            if hasattr(c_, "counters"):
                e["library_collectives_so_far"] = list(c_.counters())
            return e
        transport_runs.append(record(transports[0], comm, elapsed))
        for kind in transports[1:]:
            c2 = make_comm(kind)
            if c2 is None:
                transport_runs.append({"transport": kind, "ok": False, "error": "could not be created (see stderr)"})
                continue
            comm = c2
            transport_runs.append(record(kind, comm, timed_region()))
        good = [e for e in transport_runs if e.get("ok")]
        if not good:
            raise SystemExit("no transport produced a consistent product: %s" % transport_runs)
        best = min(good, key=lambda e: e["step_us"])
        if best["transport"] != transport_runs[-1]["transport"] or not transport_runs[-1].get("ok"):
            comm = make_comm(best["transport"])   # the rest of the run (phases, parity) on the transport `value` is quoted on
            if comm is None:
                raise SystemExit("transport %s could not be re-created" % best["transport"])
        elapsed = best["step_us"] * args.steps / 1e6
    if args.kernel_only:   # child of --pmc-traffic: the launches above and the STPCG loop are all a counter pass needs
        if world == 1 and args.op == "hvp":
            h, v, _, _, _ = stpcg_setup(P, dm, p, x_gt)
            stpcg_run(h, v, 8)
            stpcg_run(h, v, 60)
        os.write(json_fd, b"{}\n")
        return
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline: the kernel alone, HIP events on the stream it runs on ------
    # (N > 1: the exchange is part of the step, so the kernel alone is timed on a handle-local product: the operand's
    # remote rows are current after the timed region above)
    ex_fn = None
    if world > 1:
        if hasattr(comm, "local_products"):
            comm.local_products(True)
        else:
            ex_fn = comm.exchange
            comm.exchange = lambda ptr, ld_: None
    for _ in range(20):
        step()
    ctx.sync()
    reps = 500
    ctx.timer_start()
    for _ in range(reps):
        step()
    kernel_us = ctx.timer_stop_ms() * 1e3 / reps
    if world > 1:
        if hasattr(comm, "local_products"):
            comm.local_products(False)
        else:
            comm.exchange = ex_fn
    if args.op == "cert":
        b_spmm, _ = algorithmic_bytes(dm["d"], dm["n"], dm["r"], dm["N"], dm["nnz"], k_op)
        b_hvp = b_spmm + (dm["n"] * dm["d"] ** 2 + dm["r"]) * 8   # + the Lambda blocks
    stats = ctx.format_stats()
    # what the FORMAT makes a product move at least: its own bytes of Q (not the CSR's 12 nnz + 4 (N + 1)), the operand
    # read once, the result written once, and for the epilogues the point's rows and the Lambda blocks
    fb = ctx.format_bytes()
    k_cols = p if args.op == "hvp" else k_op
    rows_loc = stats["local_rows"]
    comp_read = fb["total"] + 8 * rows_loc * k_cols + 8 * (dm["n"] * dm["d"] ** 2 + dm["r"]) + \
        (8 * (dm["d"] * dm["n"] + dm["r"]) * p if args.op == "hvp" else 0)
    comp_write = 8 * rows_loc * k_cols
    local_frac = stats["local_nnz"] / max(dm["nnz"], 1)
    achieved = b_hvp * local_frac / kernel_us / 1e3  # GB/s, this rank's share of the bytes
    epi = 2 if args.op == "hvp" else 1
    k_b2b = "k_spmm<%d, 3, %d>" % (ld, epi)       # the back-to-back launches
    k_loop = "k_spmm<%d, 3, 3>" % ld               # the same product inside the STPCG loop (EPI_HVP_K: + the kappa partials)
    pmc = pmc_traffic(args, ld, epi) if (world == 1 and args.pmc_traffic == "auto") else None
    kt = rocprof_kernel_times(args, ld) if (world == 1 and args.pmc_traffic == "auto" and args.op == "hvp") else None
    traffic, traffic_source = None, None
    if pmc and k_b2b in pmc:
        traffic, traffic_source = pmc[k_b2b]["read"] + pmc[k_b2b]["write"], PMC_HOW
    tpath = os.path.join(ROOT, "profiles", "hvp_traffic.json")
    if traffic is None and world == 1 and n == 100000 and p == 5 and args.op == "hvp" and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            traffic_source = ("profiles/hvp_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                              "an earlier run (FETCH_SIZE x 2 on gfx950), not collected in this run")
        except Exception:
            traffic = None

    # ---- the same kernel with its working set forced out of the 256 MiB Infinity Cache: five independent copies
    # of the problem (Q + point + operand + result, ~150 MB each) visited round-robin on one stream
    hbm_us, hbm_spmm_us, chunk_us = None, None, None
    if world == 1:
        # per-chunk statistics of the headline (median and p10 / p90): 30 chunks of 100 back-to-back products
        chunk_us = []
        for _ in range(30):
            ctx.timer_start()
            for _ in range(100):
                step()
            chunk_us.append(ctx.timer_stop_ms() * 1e3 / 100)
        chunk_us.sort()
        others = []
        for _ in range(4):
            c2 = capi.Context(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], rowptr, colidx, vals, device=local_rank)
            c2.set_rank(p)
            c2.set_stream(stream.cuda_stream)
            y2, x2, o2 = y.clone(), x.clone(), torch.zeros_like(out)
            c2.set_point_dev(y2.data_ptr())
            xk2 = ok2 = None
            if args.op == "cert":
                xk2, ok2 = xk.clone(), torch.zeros_like(ok)
            others.append((c2, x2, o2, xk2, ok2))
        ring = [(ctx, x, out, xk if args.op == "cert" else None, ok if args.op == "cert" else None)] + others

        def rotated(fn):
            for _ in range(4):
                for e in ring:
                    fn(e)
            ctx.sync()
            rounds = 60
            ctx.timer_start()
            for _ in range(rounds):
                for e in ring:
                    fn(e)
            return ctx.timer_stop_ms() * 1e3 / (rounds * len(ring))
        if args.op == "hvp":
            hbm_us = rotated(lambda e: e[0].hvp_dev(e[1].data_ptr(), e[2].data_ptr()))
            hbm_spmm_us = rotated(lambda e: e[0].spmm_dev(e[1].data_ptr(), p, e[2].data_ptr()))
        else:
            hbm_us = rotated(lambda e: e[0].certificate_product_dev(e[3].data_ptr(), k_op, e[4].data_ptr()))
        del ring, others

    # N > 1: what a product costs where, and what crosses the ranks (every rank calls; rank 0 prints)
    phases, comm_counts = None, None
    if dist is not None and args.op == "hvp":
        if hasattr(comm, "product_phases"):
            try:
                phases = comm.product_phases(x.data_ptr(), out.data_ptr(), epi=2, reps=50)
            except Exception as e:  # noqa: BLE001 -- a diagnostic, not the measurement
                phases = {"error": str(e)}
        if hasattr(comm, "counters"):
            c0 = comm.counters()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            c1 = comm.counters()
            comm_counts = {"allgathers_per_product": (c1[0] - c0[0]) / 10.0, "allreduces_per_product": (c1[1] - c0[1]) / 10.0,
                           "launches_per_product": 3 if hasattr(comm, "status") else 4,
                           "launches": ("long-row chunks | exchange (one kernel: exported rows read from X, pushed with the slots "
                                        "into the peers' mailboxes, hand-over, wait, unpack with the long rows summed in rank "
                                        "order) | slices" if hasattr(comm, "status") else
                                        "pack (+ zeroed slots) | long-row chunks | [collective] | unpack (+ long rows summed in "
                                        "rank order) | slices"),
                           "per_stpcg_iteration": "1 all-gather (the product's) + 2 all-reduces (kappa | <r,r> and <r,v>)"}
    # N > 1: one product gathered on every rank (download is collective) for the parity check on rank 0
    gathered = None
    if dist is not None:
        ctx.hvp_dev(x.data_ptr(), out.data_ptr())
        gathered = ctx.download(out.data_ptr(), p)
        torch.cuda.synchronize()

    # N = 1, the headline op: the solver's inner loop around the product (extras), timed before the result is put together
    # because the roofline of the line is the product INSIDE that loop ("steady state inside the STPCG loop")
    extras = None
    if rank == 0 and world == 1 and args.op == "hvp":
        extras = solver_loop_timings(P, ctx, dm, p, x, out, kernel_us, x_gt)

    result = None
    if rank == 0:
        kname = "cora::k_spmm<%d, 3, %d> (LD=%d, d=3, %s)" % (ld, epi, ld, "EPI_HVP" if args.op == "hvp" else "EPI_S")
        b2b = {
            "bound": "hbm",
            "kernel": kname,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "compulsory_bytes": comp_read + comp_write,
            "compulsory_read_bytes": comp_read,
            "compulsory_write_bytes": comp_write,
            "format_bytes": fb,
            "traffic_read": pmc[k_b2b]["read"] if pmc and k_b2b in pmc else None,
            "traffic_write": pmc[k_b2b]["write"] if pmc and k_b2b in pmc else None,
            "kernel_us": kernel_us,
            "bytes_per_launch": b_hvp * local_frac,
            "how": "500 back-to-back launches between two HIP events on the handle's stream: Q and the three vectors stay in "
                   "the 256 MiB Infinity Cache (see roofline_hbm for the same kernel with the working set rotated out of it)",
        }
        if b2b["traffic_read"]:
            b2b["read_over_compulsory"] = b2b["traffic_read"] / comp_read
            b2b["write_over_compulsory"] = b2b["traffic_write"] / comp_write
        # `frac` is quoted on the ALGORITHMIC bytes (the reference's CSR, 12 B per entry), which the device format
        # does not move (25.6 MB instead of 56.4 MB at 10^5 poses).  Beside it, the same kernel time against what the kernel
        # really moved (PMC) and against the least the format could move (round-5 review: "report the Hvp against what it moves")
        if traffic:
            b2b["frac_traffic"] = traffic / kernel_us / 1e3 / HBM_PEAK_GBS
        b2b["frac_compulsory"] = (comp_read + comp_write) / kernel_us / 1e3 / HBM_PEAK_GBS
        result = {
            "metric": "riemannian_hessian_vector_products_per_sec" if args.op == "hvp" else
                      "certificate_operator_products_per_sec",
            "value": args.steps / elapsed,
            "unit": "Hvp/s" if args.op == "hvp" else "products/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic SE(3) odometry chain, %d poses + %d landmarks + %d range edges (seed 42); "
                            "%s at relaxation rank p=%d; N=%d, nnz(Q)=%d"
                            % (dm["n"], dm["l"], dm["r"],
                               "Hvp = Proj_Y((Q - Lambda) Ydot)" if args.op == "hvp" else
                               "certificate operator (Q - Lambda) X, %d columns" % k_op, p, dm["N"], dm["nnz"]),
                "parallelism": "1 GPU" if world == 1 else
                               "rows of Q over %d GPUs (pose-aligned, nnz-balanced) + one RCCL all-gather (issued by the library "
                               "itself on the handle's stream) of the %d rows of the operand (of %d) that other ranks read, "
                               "packed and scattered by the library's own kernels" % (world, comm.exchanged_rows, rows),
                "algorithmic_bytes_per_hvp": b_hvp,
                "algorithmic_bytes_per_spmm": b_spmm,
            },
            "value_is": "steps / wall time of the timed region (K back-to-back products between two barriers + "
                        "synchronisations, host clock); per-chunk statistics of the same launches: value_stats",
        }
        if extras is not None and extras.get("hvp_in_stpcg_us"):
            # the line's roofline: the product as the solver runs it -- inside the STPCG loop, the preconditioner's two
            # sweeps over the factor between two products -- HIP events around it in every iteration
            # (without a rocprofv3 child: the interval between two events, the event's own cost INCLUDED -- an upper bound of
            # the kernel's duration, never the flattering "minus two events in a row", which undershoots it)
            loop_us = extras["hvp_in_stpcg_with_event_us"]
            loop_src = "HIP events around the launch in every iteration, one event's cost on the stream included (upper bound; no rocprofv3 child in this run)"
            if kt and k_loop in kt:   # the kernel alone, as rocprofv3 sees it in a child run of this command
                loop_us = kt[k_loop]["us"]
                loop_src = ("rocprofv3 --kernel-trace --stats of a short child run of this command: average over %d launches of "
                            "%s inside the loop" % (kt[k_loop]["calls"], k_loop))
            rl = {
                "bound": "hbm",
                "kernel": "cora::k_spmm<%d, 3, 3> (LD=%d, d=3, EPI_HVP_K: the Hvp with the partial sums of <p, Hp>)" % (ld, ld),
                "achieved": b_hvp / loop_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": b_hvp / loop_us / 1e3 / HBM_PEAK_GBS,
                "traffic": (pmc[k_loop]["read"] + pmc[k_loop]["write"]) if pmc and k_loop in pmc else None,
                "traffic_source": PMC_HOW if pmc and k_loop in pmc else None,
                "traffic_read": pmc[k_loop]["read"] if pmc and k_loop in pmc else None,
                "traffic_write": pmc[k_loop]["write"] if pmc and k_loop in pmc else None,
                "kernel_us": loop_us, "bytes_per_launch": b_hvp, "samples": extras["hvp_in_stpcg_samples"],
                "compulsory_bytes": comp_read + comp_write,
                "kernel_us_source": loop_src,
                "kernel_us_events": {"interval": extras["hvp_in_stpcg_with_event_us"], "two_events_in_a_row": extras["event_overhead_us"],
                                     "net": extras["hvp_in_stpcg_us"]},
                "how": "the Hessian-vector product INSIDE the device-resident STPCG loop (Cholesky preconditioner: two sweeps "
                       "over the 58 MB factor and eleven vector passes between two products).  An event recorded between two "
                       "kernels of the loop costs the stream 2-5 us -- more than the boundary it sits on -- so an interval "
                       "between two events overstates the kernel (round 4's 21 us) and subtracting two events in a row "
                       "understates it: kernel_us is the kernel's own duration from a rocprofv3 kernel trace of the same loop "
                       "(profiles/r06_bench_kernel_stats.csv holds the same population), the event figures stay in "
                       "kernel_us_events",
            }
            if rl["traffic"]:
                rl["frac_traffic"] = rl["traffic"] / loop_us / 1e3 / HBM_PEAK_GBS
                rl["read_over_compulsory"] = rl["traffic_read"] / comp_read
                rl["write_over_compulsory"] = rl["traffic_write"] / comp_write
            rl["frac_compulsory"] = (comp_read + comp_write) / loop_us / 1e3 / HBM_PEAK_GBS
            rl["frac_is"] = ("frac: algorithmic bytes (the reference's CSR, SURVEY 8d) / kernel time / peak; frac_traffic: PMC bytes of "
                             "the same kernel / time / peak -- what the kernel really moved; frac_compulsory: the least the device "
                             "format can move (its own bytes once + every vector row once) / time / peak")
            result["roofline"] = rl
            result["roofline_cache"] = b2b
        else:
            result["roofline"] = b2b
        if extras is not None and extras.get("stpcg_phase_us") and extras["stpcg_phase_us"].get("forward_sweep"):
            ent = extras.pop("_stpcg_entries")
            ph = extras["stpcg_phase_us"]
            ab = stpcg_bytes(dm, p, extras["preconditioner"]["nnz_L"], ent)
            kmap = {"product": k_loop, "kappa": "k_kappa_finish", "forward_sweep": "k_subblock<%d, false, 1>" % ld,
                    "top_forward": "k_rowop<%d>" % ld, "top_backward": "k_rowop<%d>" % ld,
                    "backward_sweep": "k_subblock<%d, true, 3>" % ld}
            kern = {}
            ov = ph.get("event_overhead") or 0.0   # two events in a row: what every figure below carries
            for name in ("product", "kappa", "forward_sweep", "top_forward", "top_backward", "backward_sweep"):
                if name == "kappa" and ph.get("kappa_folded"):
                    kern[name] = {"kernel": None, "us": None, "note": "no launch: every block of the forward sweep adds the "
                                  "product's partial sums behind the loads of its right-hand sides"}
                    continue
                # (the last stage's two products are two launches of ONE kernel: rocprofv3's average cannot tell them apart,
                # so they keep the event interval minus two events in a row; every other launch, without a rocprofv3 child,
                # the interval with the event's cost included -- an upper bound)
                net = (max(ph[name] - ov, 0.0) if name.startswith("top_") else ph[name]) if ph.get(name) else None
                if kt and kmap[name] in kt and not name.startswith("top_"):
                    net = kt[kmap[name]]["us"]          # the kernel's own duration (rocprofv3 child run)
                e = {"kernel": kmap[name], "us": net, "us_event_interval": ph.get(name), "algorithmic_bytes": ab[name]}
                if net and ab[name]:
                    e["frac"] = ab[name] / net / 1e3 / HBM_PEAK_GBS
                if pmc and kmap[name] in pmc and not name.startswith("top_"):
                    e["pmc_bytes"] = pmc[kmap[name]]["read"] + pmc[kmap[name]]["write"]
                    e["pmc_read"], e["pmc_write"] = pmc[kmap[name]]["read"], pmc[kmap[name]]["write"]
                    if ab[name]:
                        e["pmc_over_algorithmic"] = e["pmc_bytes"] / ab[name]
                kern[name] = e
            if kt and kmap["top_forward"] in kt:     # the two products of the last stage are launches of one kernel
                kern["top_forward"]["us_both_products_rocprof"] = 2 * kt[kmap["top_forward"]]["us"]
            if pmc and kmap["top_forward"] in pmc:   # the two products of the last stage are launches of one kernel
                t = pmc[kmap["top_forward"]]
                kern["top_forward"]["pmc_bytes_both_products"] = 2 * (t["read"] + t["write"])
            tot_b = sum(ab.values())
            it_us = extras["stpcg_iteration_us"]
            result["roofline_stpcg"] = {
                "bound": "hbm", "what": "one sweep-fused STPCG iteration with the RegularizedCholesky preconditioner "
                                         "(src/CORA.cpp:71-92,119-122; src/CORA_preconditioners.cpp:46-83)",
                "algorithmic_bytes": tot_b, "us": it_us, "achieved": tot_b / it_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": tot_b / it_us / 1e3 / HBM_PEAK_GBS,
                "us_is": "host clock over %d iterations of cora_stpcg_warm_dev, the entry TNT calls (no events inside); over 240 "
                         "iterations: %.1f us" % (extras["stpcg_iterations_timed"], extras.get("stpcg_iteration_us_240", float("nan"))),
                "sum_of_launches_us": sum(k["us"] for k in kern.values() if k.get("us")),
                "event_overhead_us": ov,
                "launches": sum(1 for k in kern.values() if k.get("us")),
                "kernels": kern,
                "entries": ent,
                "bytes_are": "12 B per stored entry of L / of the last stage's products, 8 N p per vector pass: forward sweep L + 4 "
                             "passes (r, Hp in; r, y out), backward sweep L + 5 passes (y, p, s in; p, s out) + the point's rows; "
                             "kernels[*].us: the kernel's own average duration from a rocprofv3 kernel trace of a child run of this "
                             "command (the last stage's two products: HIP-event intervals minus two events in a row -- they are two "
                             "launches of one kernel; their rocprofv3 sum is us_both_products_rocprof); us_event_interval: HIP "
                             "events around the launch, the event's own cost included",
            }
        elif extras is not None:
            extras.pop("_stpcg_entries", None)
        if world > 1:
            result["multi_gpu"] = {
                "transport": type(comm).__name__, "transports_timed": transport_runs,
                "value_is_quoted_on": "the faster of the transports whose product is consistent (transports_timed[*].ok)",
                "exchanged_rows_per_product": getattr(comm, "exchanged_rows", None),
                "rccl_ranks": getattr(comm, "nranks", None),
                "kernel_only_us": kernel_us, "step_us": elapsed / args.steps * 1e6,
                "phases_us": phases, "collectives": comm_counts,
                "note": "phases: HIP events on the handle's stream of rank 0, serial order (the interior / boundary overlap "
                        "is off below 2 048 interior slices per rank); kernel_only_us: the same product with every "
                        "collective step skipped (cora_debug_local_products)"}
        if hbm_us is not None:
            result["roofline_hbm"] = {
                "bound": "hbm", "kernel": kname, "kernel_us": hbm_us,
                "achieved": b_hvp / hbm_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": b_hvp / hbm_us / 1e3 / HBM_PEAK_GBS,
                "how": "five independent copies of the problem visited round-robin (working set ~750 MB)",
            }
        if hbm_spmm_us is not None:   # the north star's named kernel, Q . X without an epilogue, the same way
            result["roofline_hbm_spmm"] = {
                "bound": "hbm", "kernel": "cora::k_spmm<%d, 3, 0> (EPI_NONE)" % ld, "kernel_us": hbm_spmm_us,
                "achieved": b_spmm / hbm_spmm_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": b_spmm / hbm_spmm_us / 1e3 / HBM_PEAK_GBS, "bytes_per_launch": b_spmm,
                "how": "five independent copies of the problem visited round-robin (working set ~750 MB)",
            }
        if chunk_us:
            n_c = len(chunk_us)
            result["value_stats"] = {
                "unit": result["unit"], "p10": 1e6 / chunk_us[int(0.9 * (n_c - 1))], "p50": 1e6 / chunk_us[n_c // 2],
                "p90": 1e6 / chunk_us[int(0.1 * (n_c - 1))],
                "how": "30 chunks of 100 back-to-back products after the timed region, HIP events per chunk (a short timed "
                       "region carries the launch ramp; these do not)"}
        if world > 1 and args.op == "hvp":
            # parity of the sharded product against the CPU oracle on the same operands
            from oracle import oracle as orc
            Qo = orc.CSR(rowptr, colidx, vals, dm["N"])
            dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
            Yc = orc.project_manifold(dims, Yh)
            Vc = orc.tangent_proj(dims, Yc, Vh)
            ref = orc.hvp(Qo, dims, Yc, orc.egrad(Qo, Yc), Vc)
            result["parity_max_rel_err_vs_cpu"] = float(np.abs(gathered - ref).max() / np.abs(ref).max())
        if world == 1 and args.op == "cert":
            from oracle import oracle as orc
            Qo = orc.CSR(rowptr, colidx, vals, dm["N"])
            dims = orc.Dims(dm["d"], dm["n"], dm["r"], dm["N"])
            Yc = ctx.download(y.data_ptr(), p)
            Xc = ctx.download(xk.data_ptr(), k_op)
            Lst, lob = orc.lambda_blocks(Qo, dims, Yc)
            t0c = time.perf_counter()
            ref = orc.S_apply(Qo, dims, Lst, lob, Xc)
            t_cpu = time.perf_counter() - t0c
            got = ctx.download(ok.data_ptr(), k_op)
            result["parity_max_rel_err_vs_cpu"] = float(np.abs(got - ref).max() / np.abs(ref).max())
            result["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "products/s", "cores": 1, "kind": "port",
                                      "sample": "1 product of the same workload with oracle/cora_oracle.c (single thread)"}
        if world == 1 and args.op == "hvp":
            cores = os.cpu_count()
            hv_s, reps_cpu, (Yc, Vc, ref) = cpu_baseline(rowptr, colidx, vals, dm, p, args.cpu_seconds)
            # parity of the timed GPU path against the CPU result on the same inputs
            ctx.upload(Yc, y.data_ptr())
            ctx.set_point_dev(y.data_ptr())
            ctx.upload(Vc, x.data_ptr())
            ctx.hvp_dev(x.data_ptr(), out.data_ptr())
            got = ctx.download(out.data_ptr(), p)
            result["parity_max_rel_err_vs_cpu"] = float(np.abs(got - ref).max() / np.abs(ref).max())
            result["extras"] = extras
            all_cores = None
            if not args.no_cpu_all_cores:
                # all-core column of BASELINE.md section 4: the same oracle loops under OpenMP.  A container may see
                # more logical cores than it may use, so a few thread counts are tried and the best one is reported.
                # (round 6: in the default run too -- two thread counts, about a second each -- so that the driver's record
                # carries the column; --cpu-all-cores sweeps more counts)
                best = None
                quota = cpu_quota()   # what the container may use (cgroup cpu.max), not what it sees
                counts = (4, 8, 16, 32, 64, 128, cores) if args.cpu_all_cores else (min(cores, quota), max(1, min(cores, quota) // 2))
                for th in sorted({max(1, min(t, cores, quota)) for t in counts}):
                    hv, reps_th, _ = cpu_baseline(rowptr, colidx, vals, dm, p, 1.0, threads=th)
                    if best is None or hv > best[0]:
                        best = (hv, th, reps_th)
                all_cores = result["extras"]["cpu_all_cores"] = {
                    "value": best[0], "unit": "Hvp/s", "cores": best[1],
                    "sample": "%d products, same oracle code with OpenMP row-parallel loops, threads bound to cores "
                              "(spread over the sockets), Q and the vectors first touched by the threads that use them; best "
                              "of several thread counts up to the container's CPU quota of %d (cgroup cpu.max; the host "
                              "reports %d logical cores)" % (best[2], quota, cores)}
            result["cpu_baseline"] = {
                "value": hv_s,
                "unit": "Hvp/s",
                "cores": 1,
                "kind": "port",
                "sample": "%d Hessian-vector products of the same 10^5-pose workload with oracle/cora_oracle.c "
                          "(single thread, like the reference; host has %s logical cores)" % (reps_cpu, cores),
            }
            if all_cores:   # BASELINE.md section 4's second column beside the reference-equivalent single thread
                result["cpu_baseline"]["all_cores"] = {k: all_cores[k] for k in ("value", "unit", "cores")}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
