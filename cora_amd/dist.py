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
