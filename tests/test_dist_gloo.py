"""CPU, world_size 2 (gloo): the multi-GPU combine — shard ownership, the integer-SUM all-reduce of the
zero-padded exchange buffer, the canonical host fold — gives bit-identical results to a single rank.
The per-octant partial states are produced by the same numpy routine on every rank (on GPUs they come
from the fused kernel + octant fold; the exchange + fold code under test is the product's)."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, mod

LANE_OPS = [1, 0, 0, 2, 3, 4]  # rows (add i64), two f64 sums, min i64, max i64, error flag (max u64)


def octant_partials(lo, hi, x, y):
    """Emulated per-octant state over rows [lo, hi): what K_main + K_fold leave in one exchange row."""
    out = np.zeros(len(LANE_OPS), dtype=np.uint64)
    xs, ys = x[lo:hi], y[lo:hi]
    out[0] = np.uint64(hi - lo)
    out[1] = np.float64(xs.sum()).view(np.uint64)
    out[2] = np.float64((xs * ys).sum()).view(np.uint64)  # negative sums: bit patterns with the top bit set
    out[3] = np.int64(ys.min() if hi > lo else np.iinfo(np.int64).max).view(np.uint64)
    out[4] = np.int64(ys.max() if hi > lo else np.iinfo(np.int64).min).view(np.uint64)
    out[5] = 0
    return out


def make_data(n):
    rng = np.random.default_rng(7)
    return rng.normal(size=n) * 1e6 - 3e5, rng.integers(-10**12, 10**12, size=n)


def exchange_for_rank(dist_mod, lib, rank, world, chunk_rows, x, y):
    begin, owner = dist_mod.shard_layout(lib, len(chunk_rows), world)
    row_of_chunk = np.concatenate([[0], np.cumsum(chunk_rows)])
    ex = np.zeros((8, len(LANE_OPS)), dtype=np.uint64)
    for o in range(8):
        if owner[o] == rank:
            ex[o] = octant_partials(int(row_of_chunk[begin[o]]), int(row_of_chunk[begin[o + 1]]), x, y)
    return ex


def _worker(rank, world, port, chunk_rows, n, out_path):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = mod("runtime").lib()
    dmod = mod("dist")
    x, y = make_data(n)
    ex = exchange_for_rank(dmod, lib, rank, world, chunk_rows, x, y)
    t = torch.from_numpy(ex.view(np.int64).reshape(-1).copy())
    dmod.all_reduce_exchange(dist, t)  # the query's single collective
    state = dmod.fold_exchange(lib, t.numpy().view(np.uint64).reshape(8, -1), LANE_OPS)
    if rank == 0:
        np.save(out_path, state)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2])
def test_two_rank_combine_is_bit_identical_to_one_rank(tmp_path, world):
    import torch.multiprocessing as mp

    chunk_rows = [1000] * 37 + [123]  # ragged last chunk, octants of unequal size
    n = sum(chunk_rows)
    out = str(tmp_path / "state.npy")
    mp.spawn(_worker, args=(world, _free_port(), chunk_rows, n, out), nprocs=world, join=True)
    got = np.load(out)

    lib = mod("runtime").lib()
    dmod = mod("dist")
    x, y = make_data(n)
    want = dmod.fold_exchange(lib, exchange_for_rank(dmod, lib, 0, 1, chunk_rows, x, y), LANE_OPS)
    assert got.tobytes() == want.tobytes()  # bit-exact, f64 lanes included
    assert int(want[0]) == n
    assert want[3].view(np.int64) == y.min() and want[4].view(np.int64) == y.max()
    # and the fold really is "octants in order": a plain left-to-right sum of the 8 octant sums
    ex = exchange_for_rank(dmod, lib, 0, 1, chunk_rows, x, y)
    acc = 0.0
    for o in range(8):
        acc += float(ex[o, 1:2].view(np.float64)[0])
    assert np.float64(acc).view(np.uint64) == want[1]


# ---------------------------------------------------------------------------------------------------
# Sharded join → GROUP BY → top-k: the product's collective driver (dist.join_groupby_topk)
# and host pieces (fold_straddlers, merge) on two gloo ranks.  The device phases (probe, per-group sums) are
# stood in by numpy with the same contract: per-group row counts, straddler pairs in row order, candidates.
# ---------------------------------------------------------------------------------------------------
class _HostJoinAgg:
    def __init__(self, groups, values, n_groups, payload):
        self.groups, self.values, self.n_payload, self.payload = groups, values, 1, payload
        self.local = np.bincount(groups, minlength=n_groups).astype(np.int64)
        self.counts = self.local.copy()

    def counts_buffer(self):
        return self.counts, len(self.counts)

    def straddlers(self):
        m = (self.local != self.counts)[self.groups]
        return self.groups[m].astype(np.uint32), self.values[m]

    def candidates(self, folded, rank, limit):
        sums, cnt = {}, {}
        for g, v in zip(self.groups, self.values):  # left-to-right f64 adds from 0.0, row order
            if self.local[g] == self.counts[g]:
                sums[g] = sums.get(g, 0.0) + float(v)
                cnt[g] = cnt.get(g, 0) + 1
        for g, s, c, r in zip(*folded):
            if r == rank:
                sums[int(g)], cnt[int(g)] = float(s), int(c)
        rows = sorted(((10_000 + g, s, cnt[g], (int(self.payload[g]), 0, 0, 0), g) for g, s in sums.items()), key=lambda t: (-t[1], t[3][0], t[4]))
        return rows[:limit], len(rows)


def _join_data(n_rows, n_groups, clustered):
    rng = np.random.default_rng(11)
    groups = np.sort(rng.integers(0, n_groups, size=n_rows)) if clustered else rng.integers(0, n_groups, size=n_rows)
    values = rng.normal(size=n_rows) * 1e5
    payload = rng.integers(0, 50, size=n_groups)
    return groups.astype(np.int64), values, payload


def _join_worker(rank, world, port, n_rows, n_groups, clustered, out_path):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt, dmod = mod("runtime"), mod("dist")
    rt.lib()
    groups, values, payload = _join_data(n_rows, n_groups, clustered)
    lo, hi = rank * n_rows // world, (rank + 1) * n_rows // world
    ja = _HostJoinAgg(groups[lo:hi], values[lo:hi], n_groups, payload)

    def all_reduce_counts(buf, n):
        dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM)

    rows, total = dmod.join_groupby_topk(dist, rt, ja, rank, world, 10, all_reduce_counts)
    if rank == 1:  # every rank holds the same answer; check a non-zero one
        np.save(out_path, np.array([(r[0], np.float64(r[1]).view(np.uint64), r[2], r[3]) for r in rows] + [(total, 0, 0, 0)], dtype=np.uint64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("clustered", [True, False])
def test_two_rank_join_groupby_topk_is_bit_identical_to_one_rank(tmp_path, clustered):
    import torch.multiprocessing as mp

    n_rows, n_groups = 20_000, 700
    out = str(tmp_path / "rows.npy")
    mp.spawn(_join_worker, args=(2, _free_port(), n_rows, n_groups, clustered, out), nprocs=2, join=True)
    got = np.load(out)
    groups, values, payload = _join_data(n_rows, n_groups, clustered)
    sums, cnt = np.zeros(n_groups), np.zeros(n_groups, dtype=np.int64)
    for g, v in zip(groups, values):  # the reference's order: one sequential fold over the whole table
        sums[g] = sums[g] + v if cnt[g] else 0.0 + v
        cnt[g] += 1
    present = [g for g in range(n_groups) if cnt[g]]
    want = sorted(present, key=lambda g: (-sums[g], payload[g], g))[:10]
    assert int(got[-1, 0]) == len(present)
    assert [int(r[0]) for r in got[:-1]] == [10_000 + g for g in want]
    assert [int(r[1]) for r in got[:-1]] == [int(np.float64(sums[g]).view(np.uint64)) for g in want]  # bit-exact f64 sums
    assert [int(r[2]) for r in got[:-1]] == [int(cnt[g]) for g in want]


# ---------------------------------------------------------------------------------------------------
# Table-wide column statistics for sharded tables (dist.share_column_stats): every rank must install the same
# (min, max), or the ranks would lower different plans.
# ---------------------------------------------------------------------------------------------------
class _StatsTable:
    def __init__(self, local):
        self.local, self.installed = local, {}

    def local_column_stats(self, f):
        return self.local.get(f)

    def set_column_stats(self, f, lo, hi):
        self.installed[f] = (lo, hi)


def _stats_worker(rank, world, port, out_path):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dmod = mod("dist")
    local = [{4: (1, 30), 10: (8035, 9000), 6: None}, {4: (5, 50), 10: (8900, 10471), 6: None}][rank]
    t = _StatsTable(local)
    dmod.share_column_stats(dist, t, [4, 10, 6], world)
    np.save(out_path + f".{rank}.npy", np.array([t.installed.get(4, (0, 0)), t.installed.get(10, (0, 0)), t.installed.get(6, (-1, -1))]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_install_identical_table_wide_statistics(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "stats")
    mp.spawn(_stats_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    assert a.tolist() == b.tolist() == [[1, 50], [8035, 10471], [-1, -1]]  # a column without statistics stays without


class _FakeSortedQuery:
    """Stands in for a PreparedQuery of the sort-based route (no GPU here): the driver under test only moves the
    ranks' partial groups and hands them to merge_groups in rank order."""

    def __init__(self, rank):
        self.rank, self.calls, self.merged = rank, [], None

    def launch(self, stream=0):
        self.calls.append("launch")

    def finish_only(self, stream=0):
        self.calls.append("finish")

    def partial_groups(self):
        n = 3 + self.rank
        return (np.full((2, n), self.rank, np.int64), np.ones((2, n), np.uint8), np.full((n, 4), 10 * self.rank, np.uint64))

    def merge_groups(self, parts):
        self.merged = parts

    def rows(self):
        return "rows"


def _sorted_groupby_worker(rank, world, port, out_path):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q = _FakeSortedQuery(rank)
    assert mod("dist").sorted_groupby(dist, q, world) == "rows"
    assert q.calls == ["launch", "finish"] and len(q.merged) == world
    np.save(out_path + f".{rank}.npy", np.array([[p[0].shape[1], int(p[0][0, 0]), int(p[2][0, 0])] for p in q.merged]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_exchange_partial_groups_in_rank_order(tmp_path):
    """dist.sorted_groupby (GROUP BY of any cardinality over a sharded table): every rank receives every rank's partial
    groups, in rank order, and merges them itself."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "parts")
    mp.spawn(_sorted_groupby_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    assert a.tolist() == b.tolist() == [[3, 0, 0], [4, 1, 10]]


# ---------------------------------------------------------------------------------------------------
# The library's own collectives (include/llkv_hip.h "Collectives") over a host-supplied transport: two gloo ranks
# behind llkv_hip_comm_init_custom.  Host-memory entry points only — the device ones need a GPU (tests/test_gpu_*).
# ---------------------------------------------------------------------------------------------------
def _comm_worker(rank, world, port, out_path):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rt = mod("runtime")
    assert rt.comm_world() == 0
    rt.comm_init_torch(dist, rank, world)
    assert rt.comm_world() == world
    # ragged contributions, one of them empty, one larger than the all-gather's padding unit
    mine = [b"", bytes(range(256)) * 5 + b"tail"][rank]
    parts = rt.comm_all_gather_v(mine)
    ok = parts == [b"", bytes(range(256)) * 5 + b"tail"]
    parts = rt.comm_all_gather_v(bytes([rank]) * (3 + 14 * rank))
    ok = ok and parts == [bytes([0]) * 3, bytes([1]) * 17]
    # table-wide dictionary of a sharded Utf8 column: sorted union, byte order (Rust's str::cmp)
    union = rt.comm_union_strings([["N", "R", "éa"], ["A", "N", "", "Zz"]][rank])
    ok = ok and union == sorted({"N", "R", "éa", "A", "", "Zz"}, key=lambda s: s.encode())
    rt.comm_destroy()
    assert rt.comm_world() == 0
    with open(out_path + str(rank), "w") as f:
        f.write("ok" if ok else "bad")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_use_the_library_collectives_over_a_host_transport(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "res")
    mp.spawn(_comm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert [open(out + str(r)).read() for r in range(2)] == ["ok", "ok"]
