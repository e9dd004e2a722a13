"""CPU checks of the host-side logic of libcora_hip.so: the library loads and
exports the whole C ABI, and the device format (sliced-ELL + long rows + row
partition) reproduces Q*X when executed on the host (test hook)."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from cora_amd import capi
from oracle import oracle as orc
from test_oracle_golden import load
from synth import make_problem


def test_library_exports_every_declared_symbol():
    L = capi.load()
    hdr = open(os.path.join(ROOT, "include", "cora_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(cora_[a-z0-9_]+)\s*\(", hdr))
    names -= {"cora_ctx"}
    assert len(names) > 40
    for n in sorted(names):
        assert hasattr(L, n), n


def _ctx(Q, dm, rank=0, world=1):
    return capi.Context(dm.d, dm.n, dm.r, dm.n_trans, Q.rowptr, Q.col, Q.val, device=-1, rank=rank,
                        world=world)


def test_format_on_fixtures(case):
    A, Q, dm = load(case)
    ctx = _ctx(Q, dm)
    X = np.random.default_rng(0).standard_normal((dm.N, 5))
    got = ctx.debug_format_spmm_host(X)
    assert np.abs(got - orc.spmm(Q, X)).max() < 1e-12
    st = ctx.format_stats()
    assert st["local_rows"] == dm.N and st["local_nnz"] == Q.nnz


@pytest.mark.parametrize("d,loops", [(2, 0), (3, 0), (3, 40)])
def test_format_synthetic(d, loops):
    A, Q, dm = make_problem(d=d, n=700, n_landmarks=3, n_ranges=400, n_loops=loops, seed=7)
    ctx = _ctx(Q, dm)
    st = ctx.format_stats()
    assert st["long_rows"] == 3  # the landmark rows
    # (chain slices do not store what Q's symmetry gives: per pose the d x d block that couples it with its index
    # predecessor, the rotation part of its translation row -- own pose and predecessor -- and the sub-diagonal of Q33)
    assert st["padded_nnz"] + st["long_nnz"] >= Q.nnz - (dm.d * dm.d + 2 * dm.d + 1) * dm.n
    for k in (1, 5, 10):
        X = np.random.default_rng(k).standard_normal((dm.N, k))
        got = ctx.debug_format_spmm_host(X)
        ref = orc.spmm(Q, X)
        assert np.abs(got - ref).max() < 1e-9 * np.abs(ref).max()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_partition_covers_all_rows(world):
    A, Q, dm = make_problem(d=3, n=500, n_landmarks=4, n_ranges=300, n_loops=10, seed=3)
    X = np.random.default_rng(1).standard_normal((dm.N, 5))
    ref = orc.spmm(Q, X)
    total = np.zeros_like(ref)
    owned = np.zeros(dm.N, dtype=int)
    maps = []
    nnz = []
    for rank in range(world):
        ctx = _ctx(Q, dm, rank, world)
        m = ctx.row_map()
        maps.append(m)
        lo, hi = ctx.shard_begin, ctx.shard_begin + ctx.shard_rows
        mine = (m >= lo) & (m < hi)
        owned += mine
        got = ctx.debug_format_spmm_host(X)
        # distributed long rows (landmark rows, world > 1): every rank holds the partial sum over the columns it owns
        is_long = np.zeros(dm.N, dtype=bool)
        is_long[ctx.long_rows()] = True
        assert is_long.sum() == (0 if world == 1 else (np.diff(Q.rowptr)[dm.dn + dm.r:] > 96).sum())
        total[mine & ~is_long] = got[mine & ~is_long]
        total[is_long] += got[is_long]
        assert np.abs(got[~mine & ~is_long]).max(initial=0) == 0.0  # only local rows are written
        if world > 1:  # no remote row of X is read on behalf of the long rows: the exchange is the chain halo
            assert len(ctx.remote_rows()) < 0.05 * dm.N + 16 * world
        assert ctx.rows == world * ctx.shard_rows
        nnz.append(ctx.format_stats()["local_nnz"])
        # a pose's d rotation rows stay together and in order on one rank
        rot = m[:dm.dn].reshape(dm.n, dm.d)
        assert np.all(np.diff(rot, axis=1) == 1)
    assert np.all(owned == 1)
    for m in maps[1:]:
        assert np.array_equal(m, maps[0])  # all ranks agree on the layout
    assert len(set(maps[0].tolist())) == dm.N
    assert np.abs(total - ref).max() < 1e-9 * np.abs(ref).max()
    assert sum(nnz) == Q.nnz
    assert max(nnz) < 1.6 * Q.nnz / world  # nnz-balanced


def test_create_rejects_bad_input():
    A, Q, dm = load("small_ra_slam_problem")
    with pytest.raises(capi.CoraError):
        capi.Context(4, dm.n, dm.r, dm.n_trans, Q.rowptr, Q.col, Q.val, device=-1)
    bad = Q.col.copy()
    bad[0] = dm.N + 5
    with pytest.raises(capi.CoraError):
        capi.Context(dm.d, dm.n, dm.r, dm.n_trans, Q.rowptr, bad, Q.val, device=-1)
    ctx = _ctx(Q, dm)
    with pytest.raises(capi.CoraError) as e:  # compute entry points need a device: no CPU fallback
        ctx.set_rank(2)
        ctx.dataMatrixProduct(np.zeros((dm.N, 2)))
    assert e.value.code == 4
    with pytest.raises(capi.CoraError):
        ctx.set_rank(1)  # p < d
