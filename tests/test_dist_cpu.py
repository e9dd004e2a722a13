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
