"""World-size-2 gloo run (CPU) of the multi-GPU data path: row partition +
all-gather exchange + local operator + partial-sum reductions (cora_amd/dist.py).
There is no GPU here, so the local operator executes the handle's device FORMAT
on the host through the test hook; everything else is the product code path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, rows_mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cora_amd import capi, host
    from cora_amd.dist import RowShardedOperator
    from oracle import oracle as orc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        P = host.Problem.synthetic(dim=3, n_poses=300, n_landmarks=3, n_ranges=150, n_loops=5, seed=11)
        P.update()
        dm = P.dims()
        _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
        ctx = capi.Context(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], rowptr, colidx, vals, device=-1,
                           rank=rank, world=world)
        k = 5
        ld = capi.load().cora_ld_for(k)
        m = ctx.row_map().astype(np.int64)
        rows, shard = ctx.rows, ctx.shard_rows

        def local_apply(full_x, full_out):  # format executed on the host (test hook)
            xi = full_x.numpy().reshape(rows, ld)
            X = np.asfortranarray(xi[m][:, :k])
            out = ctx.debug_format_spmm_host(X)
            lr = ctx.long_rows()   # distributed long rows: partial sums, added over the ranks
            if len(lr):
                t = torch.from_numpy(np.ascontiguousarray(out[lr]))
                dist.all_reduce(t)
                out[lr] = t.numpy()
            oi = full_out.numpy().reshape(rows, ld)
            mine = (m >= rank * shard) & (m < (rank + 1) * shard)
            oi[m[mine], :k] = out[mine]

        need = ctx.remote_rows() if rows_mode else None
        op = RowShardedOperator(rows, shard, ld, rank, world, torch.device("cpu"), local_apply, needed_rows=need)
        if rows_mode:
            # rows nobody asked for are never transferred: poison them, the product must not read them
            op.full_x.fill_(float("nan"))
            assert 0 < op.exchanged_rows < rows - shard or world == 1
        rng = np.random.default_rng(3)  # same on every rank
        X = rng.standard_normal((dm["N"], k))
        xi = np.zeros((rows, ld))
        xi[m, :k] = X
        x_shard = torch.from_numpy(xi[rank * shard:(rank + 1) * shard].reshape(-1).copy())
        y_shard = op.apply(x_shard).clone()
        op.operand_shard().copy_(x_shard)
        assert torch.equal(op.apply_resident(), y_shard)
        # gather the result shards and compare with the oracle on rank 0
        full = torch.zeros(rows * ld, dtype=torch.float64)
        dist.all_gather_into_tensor(full, y_shard)
        got = full.numpy().reshape(rows, ld)[m][:, :k]
        ref = orc.spmm(orc.CSR(rowptr, colidx, vals, dm["N"]), X)
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        d_par = op.dot(x_shard, y_shard)
        d_ref = float((X * ref).sum())
        q.put((rank, err, abs(d_par - d_ref) / abs(d_ref), op.exchanged_rows, rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows_mode", [False, True])
def test_two_rank_gloo_exchange(rows_mode):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, rows_mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, derr, exchanged, rows in res:
        assert err < 1e-12, (rank, err)
        assert derr < 1e-12, (rank, derr)
        if rows_mode:
            assert exchanged < rows // 2   # only the halo and the landmark couplings travel


# ---- the injected communication of partitioned handles (cora_amd.dist.TorchComm) under gloo, world 2 and 4 ----------
def _stpcg(apply_H, apply_P, dot, g, Delta, max_iters):
    """Steihaug-Toint PCG (the loop of cora_amd/csrc/host/TNT.cpp, STPCG) on whatever vectors / inner product the
    caller supplies: the sharded run passes collective operators, the reference run plain numpy ones."""
    s = np.zeros_like(g)
    r = g.copy()
    v = apply_P(r)
    p = -v
    r_v = dot(r, v)
    sigma2, s_Mp, p_M2 = 0.0, 0.0, r_v
    for it in range(max_iters):
        Hp = apply_H(p)
        kappa = dot(p, Hp)
        alpha = r_v / kappa
        sig_next = sigma2 + 2 * alpha * s_Mp + alpha * alpha * p_M2
        if not (kappa > 0) or sig_next >= Delta * Delta:
            tau = (-s_Mp + np.sqrt(s_Mp * s_Mp + p_M2 * (Delta * Delta - sigma2))) / p_M2
            return s + tau * p, it + 1
        s = s + alpha * p
        r = r + alpha * Hp
        v = apply_P(r)
        rv_new = dot(r, v)
        beta = rv_new / r_v
        r_v = rv_new
        s_Mp = beta * (s_Mp + alpha * p_M2)
        p_M2 = r_v + beta * beta * p_M2
        sigma2 = sig_next
        p = -v + beta * p
    return s, max_iters


def _comm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from cora_amd import capi, host
    from cora_amd.dist import TorchComm
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        P = host.Problem.synthetic(dim=3, n_poses=400, n_landmarks=4, n_ranges=200, n_loops=6, seed=13)
        P.update()
        dm = P.dims()
        _, _, rowptr, colidx, vals = P.matrix("DataMatrix")
        ctx = capi.Context(dm["d"], dm["n"], dm["r"], dm["n"] + dm["l"], rowptr, colidx, vals, device=-1,
                           rank=rank, world=world)
        comm = TorchComm(ctx)           # host memory: the handle has no device, the vectors below are numpy
        k = 3
        ld = capi.load().cora_ld_for(k)
        m = ctx.row_map().astype(np.int64)
        rows, shard = ctx.rows, ctx.shard_rows
        mine = (m >= rank * shard) & (m < (rank + 1) * shard)      # API rows this rank owns
        import scipy.sparse as sp
        Qs = sp.csr_matrix((vals, colidx, rowptr), shape=(dm["N"], dm["N"]))
        dinv = 1.0 / (Qs.diagonal() + 1.0)
        buf = np.zeros((rows, ld))                                 # a resident vector in the internal layout

        def apply_H(X):                                            # (Q + I) X on the owned rows; others garbage
            buf[:] = np.nan                                        # rows nobody sends must not be read
            buf[m[mine], :k] = X[mine]
            comm.exchange(buf.ctypes.data, ld)                     # the callback the library would call
            Xfull = np.where(np.isnan(buf[m][:, :k]), 0.0, buf[m][:, :k])
            out = ctx.debug_format_spmm_host(np.asfortranarray(Xfull))
            lr = ctx.long_rows()   # distributed long rows: partial sums, added over the ranks
            if len(lr):
                t = torch.from_numpy(np.ascontiguousarray(out[lr]))
                dist.all_reduce(t)
                out[lr] = t.numpy()
            res = np.full_like(X, np.nan)
            res[mine] = out[mine] + X[mine]
            return res

        def apply_P(R):
            return R * dinv[:, None]

        def dot(A, B):
            v = (C.c_double * 1)(float((A[mine] * B[mine]).sum()))
            comm.allreduce(v, 1)
            return v[0]

        import ctypes as C
        rng = np.random.default_rng(4)
        g = rng.standard_normal((dm["N"], k))
        g_sh = np.where(mine[:, None], g, np.nan)                  # each rank only holds its rows
        res = {}
        for Delta, iters in ((1e30, 12), (0.05, 40)):
            s, its = _stpcg(apply_H, apply_P, dot, g_sh, Delta, iters)
            buf[:] = 0.0
            buf[m[mine], :k] = s[mine]
            comm.allgather(buf.ctypes.data, ld)                    # every row current on every rank
            res[Delta] = (buf[m][:, :k].copy(), its)
        # the single-process reference: same loop, plain numpy
        H1 = lambda X: Qs @ X + X
        d1 = lambda A, B: float((A * B).sum())
        errs = []
        for Delta, iters in ((1e30, 12), (0.05, 40)):
            s1, its1 = _stpcg(H1, apply_P, d1, g, Delta, iters)
            assert its1 == res[Delta][1]
            errs.append(float(np.abs(res[Delta][0] - s1).max() / np.abs(s1).max()))
        q.put((rank, max(errs), comm.exchanged_rows, rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_stpcg_through_torchcomm_matches_single_process(world):
    """Exchange (pack / one all-gather / scatter of the rows somebody reads), all-reduce and all-gather of
    cora_amd.dist.TorchComm -- the callbacks cora_set_comm installs -- drive a full Steihaug-Toint PCG solve on a
    graph cut into `world` row partitions; the step equals the single-process solve to 1e-10."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, exchanged, rows in res:
        assert err < 1e-10, (rank, err)
        assert exchanged < rows // 2
