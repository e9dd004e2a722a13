"""Multi-GPU plumbing for the row-partitioned operator.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every
rank owns `shard_rows` consecutive rows of the library's internal row order
(cora_shard_begin / cora_shard_rows); a resident vector is the concatenation of
the rank shards, so the only exchange step of a product  out = S X  is one
all-gather of X -- of whole shards, or (needed_rows given) of just the rows that
some other rank's part of Q actually reads: the halo of the pose chain and the
rows coupled to remote landmarks, typically a fifth of the vector.  Reductions
(inner products, the cost) are sums of per-rank partial results.

The local operator is passed in as a callable so that the exchange logic is
testable with the gloo backend on CPU (tests/test_dist_cpu.py)."""
import torch
import torch.distributed as dist


class RowShardedOperator:
    """out_shard = (local rows of the operator)(exchange(x_shard)).

    needed_rows: 1-D integer tensor/array of the internal rows outside this rank's shard that the local
    operator reads (cora_remote_rows).  When given, the exchange moves only the rows somebody needs:
    every rank packs the rows it owns that appear in any other rank's list, one all-gather of the packed
    buffers follows, and the received rows are scattered into the full-length operand.  Rows nobody
    asked for are never transferred (their copy on this rank is stale and unread)."""

    def __init__(self, rows, shard_rows, ld, rank, world, device, local_apply, group=None, needed_rows=None):
        assert rows == shard_rows * world
        self.rows, self.shard_rows, self.ld = rows, shard_rows, ld
        self.rank, self.world = rank, world
        self.local_apply = local_apply  # f(full_x tensor, full_out tensor): writes the local rows of out
        self.group = group
        self.device = device
        self.full_x = torch.zeros(rows * ld, dtype=torch.float64, device=device)
        self.full_out = torch.zeros(rows * ld, dtype=torch.float64, device=device)
        self.rows_mode = needed_rows is not None and world > 1
        self.exchanged_rows = rows - shard_rows if world > 1 else 0  # rows received per product
        if self.rows_mode:
            self._plan_row_exchange(torch.as_tensor(needed_rows, dtype=torch.int64).to(device))

    # ---- setup of the row exchange: who sends what, agreed on once -------------------------------------
    def _plan_row_exchange(self, need):
        world, dev = self.world, self.device
        # all ranks learn every need-list (padded to a common length with -1)
        n_local = torch.tensor([need.numel()], dtype=torch.int64, device=dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, n_local, group=self.group)
        n_max = max(int(c.item()) for c in counts)
        padded = torch.full((max(n_max, 1),), -1, dtype=torch.int64, device=dev)
        padded[:need.numel()] = need
        lists = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(lists, padded, group=self.group)
        wanted = torch.cat([l[l >= 0] for r, l in enumerate(lists) if r != self.rank]) if world > 1 else need[:0]
        # rows of MY shard that somebody else reads, ascending; every rank pads to the longest list with its
        # own first row (a duplicate that carries the same value, so scattering it twice is harmless)
        lo = self.rank * self.shard_rows
        mine = torch.unique(wanted[(wanted >= lo) & (wanted < lo + self.shard_rows)])
        m_local = torch.tensor([mine.numel()], dtype=torch.int64, device=dev)
        dist.all_gather(counts, m_local, group=self.group)
        e_max = max(1, max(int(c.item()) for c in counts))
        export = torch.full((e_max,), lo, dtype=torch.int64, device=dev)
        export[:mine.numel()] = mine
        all_exports = [torch.empty_like(export) for _ in range(world)]
        dist.all_gather(all_exports, export, group=self.group)
        self.export_idx = export                      # rows of full_x this rank sends (global row ids)
        self.recv_idx = torch.cat(all_exports)        # where the gathered rows go, rank by rank
        self.send = torch.zeros(e_max * self.ld, dtype=torch.float64, device=dev)
        self.recv = torch.zeros(world * e_max * self.ld, dtype=torch.float64, device=dev)
        self.x2d = self.full_x.view(self.rows, self.ld)
        self.exchanged_rows = world * e_max

    def shard_slice(self):
        n = self.shard_rows * self.ld
        return slice(self.rank * n, (self.rank + 1) * n)

    def exchange(self, x_shard):
        """The one data-path collective: all-gather of the operand (whole shards, or the needed rows)."""
        if self.world == 1:
            self.full_x.copy_(x_shard)
        elif not self.rows_mode:
            dist.all_gather_into_tensor(self.full_x, x_shard, group=self.group)
        else:
            self.full_x[self.shard_slice()].copy_(x_shard)
            torch.index_select(self.x2d, 0, self.export_idx, out=self.send.view(-1, self.ld))
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            self.x2d.index_copy_(0, self.recv_idx, self.recv.view(-1, self.ld))
        return self.full_x

    def operand_shard(self):
        """This rank's rows of the full-length operand: a solver that keeps its vector here skips the copy
        of apply() and calls apply_resident()."""
        return self.full_x[self.shard_slice()]

    def apply_resident(self):
        """apply() for an operand already written to operand_shard()."""
        if self.world == 1:
            self.local_apply(self.full_x, self.full_out)
            return self.full_out
        if not self.rows_mode:
            dist.all_gather_into_tensor(self.full_x, self.operand_shard().clone(), group=self.group)
        else:
            torch.index_select(self.x2d, 0, self.export_idx, out=self.send.view(-1, self.ld))
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            self.x2d.index_copy_(0, self.recv_idx, self.recv.view(-1, self.ld))
        self.local_apply(self.full_x, self.full_out)
        return self.full_out[self.shard_slice()]

    def apply(self, x_shard):
        if self.world == 1:  # the shard IS the vector: no exchange, no copy
            self.local_apply(x_shard, self.full_out)
            return self.full_out
        self.exchange(x_shard)
        self.local_apply(self.full_x, self.full_out)
        return self.full_out[self.shard_slice()]

    def dot(self, a_shard, b_shard):
        s = torch.dot(a_shard, b_shard).reshape(1)
        if self.world > 1:
            dist.all_reduce(s, group=self.group)
        return float(s.item())


# ----------------------------------------------------------------------------------------------------------
# Injected communication of a partitioned handle (include/cora_hip.h, cora_set_comm).  With one of these
# installed every resident entry point of the C ABI -- and the C++ host above it: TNT, LOBPCG, solveCORA -- is
# collective: all ranks call the same sequence.
# ----------------------------------------------------------------------------------------------------------
import ctypes as _C
import threading as _threading

import numpy as _np


class _CommBase:
    """Callback plumbing shared by the communicators: ctypes trampolines that never let an exception cross the
    C boundary (a failed step returns 1 and the library reports CORA_ERR_HIP)."""

    def _install(self, ctx):
        from . import capi
        self.ctx = ctx
        self.error = None

        def guard(f):
            def g(*a):
                try:
                    f(*a)
                    return 0
                except BaseException as e:  # noqa: BLE001 -- reported through the status code
                    self.error = e
                    return 1
            return g

        self._cb = (capi.EXCHANGE_FN(guard(lambda user, ptr, ld: self.exchange(int(ptr), int(ld)))),
                    capi.ALLREDUCE_FN(guard(lambda user, vals, n: self.allreduce(vals, int(n)))),
                    capi.ALLGATHER_FN(guard(lambda user, ptr, ld: self.allgather(int(ptr), int(ld)))))
        return self._cb


def _plan_exports(need_lists, rank, shard_rows):
    """need_lists[r]: rows rank r reads outside its shard.  Returns, for `rank`, the rows of its shard that any
    other rank reads (ascending)."""
    lo = rank * shard_rows
    wanted = [l for r, l in enumerate(need_lists) if r != rank and len(l)]
    if not wanted:
        return _np.zeros(0, dtype=_np.int64)
    w = _np.concatenate(wanted)
    return _np.unique(w[(w >= lo) & (w < lo + shard_rows)])


class TorchComm(_CommBase):
    """One rank per process, torch.distributed underneath (backend "nccl" = RCCL over xGMI; "gloo" on CPU for the
    tests).  The exchange moves only the rows somebody reads: pack (cora_pack_rows_dev) -> ONE all-gather of the
    packed rows -> scatter (cora_scatter_rows_dev); every rank pads its export list to the longest one with its own
    first row.  `device` None = the vectors live in host memory (handles without a device: format tests)."""

    def __init__(self, ctx, group=None, device=None):
        self.group, self.device = group, device
        self.rank, self.world = ctx.rank, ctx.world
        self.rows, self.shard = ctx.rows, ctx.shard_rows
        dev = device if device is not None else torch.device("cpu")
        self.dev = dev
        need = torch.as_tensor(ctx.remote_rows(), dtype=torch.int64)
        # every rank learns every need-list
        cnt = torch.tensor([need.numel()], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(cnt) for _ in range(self.world)]
        dist.all_gather(counts, cnt, group=group)
        n_max = max(1, max(int(c.item()) for c in counts))
        padded = torch.full((n_max,), -1, dtype=torch.int64, device=dev)
        padded[:need.numel()] = need.to(dev)
        lists = [torch.empty_like(padded) for _ in range(self.world)]
        dist.all_gather(lists, padded, group=group)
        need_lists = [l[l >= 0].cpu().numpy() for l in lists]
        mine = _plan_exports(need_lists, self.rank, self.shard)
        e_max = max(1, max(len(_plan_exports(need_lists, r, self.shard)) for r in range(self.world)))
        export = _np.full(e_max, self.rank * self.shard, dtype=_np.int32)
        export[:len(mine)] = mine
        all_exports = _np.concatenate([
            _np.concatenate([_plan_exports(need_lists, r, self.shard).astype(_np.int32),
                             _np.full(e_max - len(_plan_exports(need_lists, r, self.shard)), r * self.shard, _np.int32)])
            for r in range(self.world)])
        self.e_max = e_max
        self.export_idx = torch.from_numpy(export).to(dev)
        self.recv_idx = torch.from_numpy(all_exports).to(dev)
        self.exchanged_rows = self.world * e_max
        self._buf = {}
        if device is not None and device.type == "cuda":
            ctx.set_stream(torch.cuda.current_stream(device).cuda_stream)  # collectives and kernels on one stream
        self._install(ctx)
        ctx.set_comm(*self._cb)

    def _buffers(self, ld):
        if ld not in self._buf:
            self._buf[ld] = (torch.zeros(self.e_max * ld, dtype=torch.float64, device=self.dev),
                             torch.zeros(self.world * self.e_max * ld, dtype=torch.float64, device=self.dev),
                             torch.zeros(self.rows * ld, dtype=torch.float64, device=self.dev))
        return self._buf[ld]

    def _host_view(self, ptr, ld):
        return torch.from_numpy(_np.ctypeslib.as_array((_C.c_double * (self.rows * ld)).from_address(ptr))).view(self.rows, ld)

    def exchange(self, ptr, ld):
        send, recv, _ = self._buffers(ld)
        if self.dev.type == "cuda":
            self.ctx.pack_rows_dev(ptr, ld, self.export_idx.data_ptr(), self.e_max, send.data_ptr())
            dist.all_gather_into_tensor(recv, send, group=self.group)
            self.ctx.scatter_rows_dev(recv.data_ptr(), ld, self.recv_idx.data_ptr(), self.world * self.e_max, ptr)
        else:
            x = self._host_view(ptr, ld)
            torch.index_select(x, 0, self.export_idx.long(), out=send.view(-1, ld))
            dist.all_gather_into_tensor(recv, send, group=self.group)
            x.index_copy_(0, self.recv_idx.long(), recv.view(-1, ld))

    def allreduce(self, vals, n):
        t = torch.tensor([vals[i] for i in range(n)], dtype=torch.float64, device=self.dev)
        dist.all_reduce(t, group=self.group)
        out = t.cpu().tolist()
        for i in range(n):
            vals[i] = out[i]

    def allgather(self, ptr, ld):
        _, _, full = self._buffers(ld)
        n = self.shard * ld
        if self.dev.type == "cuda":
            self.ctx.copy_shard_dev(ptr, ld, self.rank, full.data_ptr())
            dist.all_gather_into_tensor(full, full[self.rank * n:(self.rank + 1) * n].clone(), group=self.group)
            for r in range(self.world):
                if r != self.rank:
                    self.ctx.copy_shard_dev(full.data_ptr(), ld, r, ptr)
        else:
            x = self._host_view(ptr, ld).view(-1)
            dist.all_gather_into_tensor(full, x[self.rank * n:(self.rank + 1) * n].clone(), group=self.group)
            x.copy_(full)


class ThreadGroup:
    """Meeting point of `world` ThreadComm ranks that live in one process and share one GPU (tests: every
    partition of a big graph on a single device)."""

    def __init__(self, world):
        self.world = world
        self.barrier = _threading.Barrier(world)
        self.ptrs = [0] * world
        self.vals = [None] * world
        self.needs = [None] * world


class ThreadComm(_CommBase):
    """Rank `ctx.rank` of a ThreadGroup.  Same contract as TorchComm; the transport is device-to-device row copies
    (cora_copy_rows_dev / cora_copy_shard_dev) between the ranks' vectors, the reductions add in rank order."""

    def __init__(self, ctx, group):
        import torch as _torch
        self.g = group
        self.rank, self.world = ctx.rank, ctx.world
        need = ctx.remote_rows()
        shard = ctx.shard_rows
        self.by_owner = []
        for r in range(self.world):
            rows = need[(need >= r * shard) & (need < (r + 1) * shard)].astype(_np.int32)
            self.by_owner.append(_torch.from_numpy(rows).cuda() if len(rows) else None)
        self.exchanged_rows = int(len(need))
        self._install(ctx)
        ctx.set_comm(*self._cb)

    def _meet(self):
        self.g.barrier.wait(timeout=600)

    def exchange(self, ptr, ld):
        self.ctx.sync()
        self.g.ptrs[self.rank] = ptr
        self._meet()
        for r, rows in enumerate(self.by_owner):
            if rows is not None and r != self.rank:
                self.ctx.copy_rows_dev(self.g.ptrs[r], ld, rows.data_ptr(), rows.numel(), ptr)
        self.ctx.sync()
        self._meet()

    def allreduce(self, vals, n):
        self.g.vals[self.rank] = [vals[i] for i in range(n)]
        self._meet()
        tot = [0.0] * n
        for r in range(self.world):
            for i in range(n):
                tot[i] += self.g.vals[r][i]
        self._meet()
        for i in range(n):
            vals[i] = tot[i]

    def allgather(self, ptr, ld):
        self.ctx.sync()
        self.g.ptrs[self.rank] = ptr
        self._meet()
        for r in range(self.world):
            if r != self.rank:
                self.ctx.copy_shard_dev(self.g.ptrs[r], ld, r, ptr)
        self.ctx.sync()
        self._meet()


# ----------------------------------------------------------------------------------------------------------
# The library's own communication (include/cora_hip.h, cora_comm_create_*): planning, pack, transport and
# scatter all run in C++ -- Python only creates the communicator.
# ----------------------------------------------------------------------------------------------------------
class NativeLocalGroup:
    """cora_local_group: meeting point of `world` ranks that are threads of this process (tests on a 1-GPU box)."""

    def __init__(self, world):
        from . import capi
        L = capi.load()
        L.cora_local_group_create.restype = _C.c_void_p
        L.cora_local_group_create.argtypes = [_C.c_int]
        L.cora_local_group_destroy.restype = None
        L.cora_local_group_destroy.argtypes = [_C.c_void_p]
        self.L, self.world = L, world
        self.h = L.cora_local_group_create(int(world))
        if not self.h:
            raise RuntimeError("cora_local_group_create failed")
        L.cora_local_group_abort.restype = None
        L.cora_local_group_abort.argtypes = [_C.c_void_p]
        self.barrier = self   # the test harness calls group.barrier.abort() when a rank fails

    def abort(self):
        self.L.cora_local_group_abort(_C.c_void_p(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.cora_local_group_destroy(_C.c_void_p(self.h))
            self.h = None


class _NativeComm:
    def _finish(self, ctx):
        L = ctx.L
        L.cora_comm_exchanged_rows.restype = _C.c_int64
        L.cora_comm_exchanged_rows.argtypes = [_C.c_void_p]
        self.ctx = ctx
        self.rank, self.world = ctx.rank, ctx.world
        self.exchanged_rows = int(L.cora_comm_exchanged_rows(ctx.h))


    def enable(self, on=True):
        self.ctx.L.cora_comm_native_enable.argtypes = [_C.c_void_p, _C.c_int]
        self.ctx._chk(self.ctx.L.cora_comm_native_enable(self.ctx.h, int(bool(on))))

    def counters(self):
        """(all-gathers, all-reduces) the library's own communication has issued on this handle so far."""
        out = (_C.c_long * 2)()
        self.ctx.L.cora_comm_counters.argtypes = [_C.c_void_p, _C.POINTER(_C.c_long)]
        self.ctx._chk(self.ctx.L.cora_comm_counters(self.ctx.h, out))
        return int(out[0]), int(out[1])

    def gathered_rows(self):
        """Rows of resident vectors received through all-gathers (whole shards or packed pieces) so far."""
        self.ctx.L.cora_comm_gathered_rows.restype = _C.c_longlong
        self.ctx.L.cora_comm_gathered_rows.argtypes = [_C.c_void_p]
        return int(self.ctx.L.cora_comm_gathered_rows(self.ctx.h))

    def product_phases(self, x_ptr, out_ptr, epi=2, reps=50):
        """Microseconds per phase of a product: pack | long-row chunks | all-gather | unpack | slices (collective call)."""
        us = (_C.c_double * 5)()
        self.ctx.L.cora_debug_product_phases.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_int, _C.c_int,
                                                         _C.POINTER(_C.c_double)]
        self.ctx._chk(self.ctx.L.cora_debug_product_phases(self.ctx.h, _C.c_void_p(x_ptr), _C.c_void_p(out_ptr), int(epi), int(reps), us))
        return dict(zip(["pack_us", "long_row_chunks_us", "allgather_us", "unpack_us", "slices_us"], [float(v) for v in us]))

    def local_products(self, on=True):
        """Timing hook: products without any collective step (the kernel alone)."""
        self.ctx.L.cora_debug_local_products.argtypes = [_C.c_void_p, _C.c_int]
        self.ctx._chk(self.ctx.L.cora_debug_local_products(self.ctx.h, int(bool(on))))

    def overlap(self, mode=1):
        """0: products run after the exchange of their operand; 1 (default): they overlap it with their interior slices
        when those are worth a launch of their own; 2: always."""
        self.ctx.L.cora_comm_overlap_enable.argtypes = [_C.c_void_p, _C.c_int]
        self.ctx._chk(self.ctx.L.cora_comm_overlap_enable(self.ctx.h, int(mode)))

    def overlap_active(self):
        self.ctx.L.cora_comm_overlap_active.argtypes = [_C.c_void_p]
        return bool(self.ctx.L.cora_comm_overlap_active(self.ctx.h))


class NativeLocalComm(_NativeComm):
    """Rank `ctx.rank` of a NativeLocalGroup: cora_comm_create_local (collective over the group's threads)."""

    def __init__(self, ctx, group):
        ctx.L.cora_comm_create_local.argtypes = [_C.c_void_p, _C.c_void_p]
        ctx._chk(ctx.L.cora_comm_create_local(ctx.h, _C.c_void_p(group.h)))
        self.group = group
        self._finish(ctx)


class NativeP2PComm(_NativeComm):
    """Device-side collectives over peer-mapped mailboxes (cora_comm_create_p2p): no RCCL, no host on the data path.
    `gather_blobs(my_blob: bytes) -> list of every rank's blob in rank order` is the launcher's part: torch.distributed's
    all_gather_object by default (any backend), or whatever the caller passes (threads of one process: a shared list)."""

    BLOB = 128

    def __init__(self, ctx, group=None, gather_blobs=None):
        L = ctx.L
        L.cora_comm_p2p_handle.argtypes = [_C.c_void_p, _C.c_void_p]
        L.cora_comm_create_p2p.argtypes = [_C.c_void_p, _C.c_void_p]
        buf = (_C.c_ubyte * self.BLOB)()
        ctx._chk(L.cora_comm_p2p_handle(ctx.h, buf))
        mine = bytes(buf)
        if gather_blobs is not None:
            blobs = gather_blobs(mine)
        elif ctx.world > 1:
            blobs = [None] * ctx.world
            dist.all_gather_object(blobs, mine, group=group)
        else:
            blobs = [mine]
        assert len(blobs) == ctx.world and all(len(b) == self.BLOB for b in blobs)
        raw = b"".join(blobs)
        ctx._chk(L.cora_comm_create_p2p(ctx.h, (_C.c_ubyte * len(raw)).from_buffer_copy(raw)))
        self._finish(ctx)

    def status(self):
        out = (_C.c_long * 6)()
        self.ctx.L.cora_comm_p2p_status.argtypes = [_C.c_void_p, _C.POINTER(_C.c_long)]
        self.ctx._chk(self.ctx.L.cora_comm_p2p_status(self.ctx.h, out))
        return dict(zip(["collectives", "kernels", "timeouts", "memory_kind", "allgathers", "allreduces"], [int(v) for v in out]))


class NativeRcclComm(_NativeComm):
    """One rank per process and GPU: RCCL called by the library itself (cora_comm_create_rccl).  The 128-byte id is
    made on rank 0 and broadcast with torch.distributed (any backend); after that torch is not on the data path."""

    def __init__(self, ctx, group=None, device=None):
        L = ctx.L
        L.cora_rccl_unique_id.argtypes = [_C.c_void_p]
        L.cora_comm_create_rccl.argtypes = [_C.c_void_p, _C.c_void_p]
        buf = (_C.c_ubyte * 128)()
        if ctx.rank == 0:
            ctx._chk(L.cora_rccl_unique_id(buf))
        if ctx.world > 1:
            dev = device if (device is not None and dist.get_backend(group) == "nccl") else torch.device("cpu")
            t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0, group=group)
            raw = bytes(t.cpu().tolist())
            buf = (_C.c_ubyte * 128).from_buffer_copy(raw)
        ctx._chk(L.cora_comm_create_rccl(ctx.h, buf))
        self._finish(ctx)
        out = (_C.c_int * 2)()
        L.cora_comm_rccl_ranks.argtypes = [_C.c_void_p, _C.POINTER(_C.c_int)]
        ctx._chk(L.cora_comm_rccl_ranks(ctx.h, out))
        self.nranks, self.user_rank = int(out[0]), int(out[1])   # ncclCommCount / ncclCommUserRank of the library's communicator
