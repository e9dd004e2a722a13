"""Multi-GPU plumbing for the row-partitioned operator.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every
rank owns `shard_rows` consecutive rows of the library's internal row order
(cora_shard_begin / cora_shard_rows); a resident vector is the concatenation of
the rank shards, so the only exchange step of a product  out = S X  is one
all-gather of X.  Reductions (inner products, the cost) are sums of per-rank
partial results.

The local operator is passed in as a callable so that the exchange logic is
testable with the gloo backend on CPU (tests/test_dist_cpu.py)."""
import torch
import torch.distributed as dist


class RowShardedOperator:
    """out_shard = (local rows of the operator)(all-gather(x_shard))."""

    def __init__(self, rows, shard_rows, ld, rank, world, device, local_apply, group=None):
        assert rows == shard_rows * world
        self.rows, self.shard_rows, self.ld = rows, shard_rows, ld
        self.rank, self.world = rank, world
        self.local_apply = local_apply  # f(full_x tensor, full_out tensor): writes the local rows of out
        self.group = group
        self.full_x = torch.zeros(rows * ld, dtype=torch.float64, device=device)
        self.full_out = torch.zeros(rows * ld, dtype=torch.float64, device=device)

    def shard_slice(self):
        n = self.shard_rows * self.ld
        return slice(self.rank * n, (self.rank + 1) * n)

    def exchange(self, x_shard):
        """The one data-path collective: all-gather of the operand shards."""
        if self.world == 1:
            self.full_x.copy_(x_shard)
        else:
            dist.all_gather_into_tensor(self.full_x, x_shard, group=self.group)
        return self.full_x

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
