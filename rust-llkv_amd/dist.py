"""Multi-GPU combine: one process per GPU, chunk-sharded tables, ONE collective per query.

The per-rank partial aggregate state lives in the library's exchange buffer as int64 lanes
([8 octants][lanes]); octants a rank does not own are zero.  An integer-SUM all-reduce of that
buffer therefore *concatenates* the ranks' states bit-exactly whatever the lane type (f64 sums
travel as their bit patterns), and the canonical host fold over the 8 octants gives the same
result for 1/2/4/8 GPUs.  The collective is torch.distributed: backend "nccl" (= RCCL over
xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

OCTANTS = 8


def shard_layout(lib, n_chunks: int, world: int):
    """(octant_chunk_begin[9], octant_owner[8]) from the C-ABI host logic."""
    begin = (C.c_uint32 * (OCTANTS + 1))()
    owner = (C.c_uint32 * OCTANTS)()
    rc = lib.llkv_hip_shard_layout(C.c_uint32(n_chunks), C.c_uint32(world), begin, owner)
    if rc != 0:
        raise ValueError(lib.llkv_hip_last_error().decode())
    return list(begin), list(owner)


def all_reduce_exchange(dist, exchange_i64, group=None):
    """The query's only collective: SUM over int64 lanes (exact concatenation)."""
    dist.all_reduce(exchange_i64, op=dist.ReduceOp.SUM, group=group)
    return exchange_i64


def fold_exchange(lib, exchange: np.ndarray, lane_ops: Sequence[int]) -> np.ndarray:
    """Canonical fold [8][lanes] → [lanes] through the library's host code."""
    lanes = len(lane_ops)
    ex = np.ascontiguousarray(exchange, dtype=np.uint64).reshape(OCTANTS * lanes)
    ops = np.asarray(lane_ops, dtype=np.uint8)
    out = np.zeros(lanes, dtype=np.uint64)
    rc = lib.llkv_hip_fold_exchange(ex.ctypes.data_as(C.c_void_p), ops.ctypes.data_as(C.c_void_p), C.c_uint32(lanes),
                                    out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(lib.llkv_hip_last_error().decode())
    return out


def table_wide_dictionary(dist, local_values, world: int) -> List[str]:
    """Sorted union of the shards' distinct Utf8 values (staging-time agreement on dictionary codes).
    ``local_values``: uint8 array of 1-byte strings or a sequence of str."""
    if isinstance(local_values, np.ndarray) and local_values.dtype == np.uint8:
        seen = np.zeros(256, dtype=bool)
        seen[local_values] = True  # one pass, no sort of the column
        local = sorted(chr(int(v)) for v in np.flatnonzero(seen))
    else:
        local = sorted(set(local_values))
    if world <= 1:
        return local
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    return sorted(set().union(*gathered))


def share_column_stats(dist, table, field_ids, world: int) -> None:
    """Table-wide integer statistics for a sharded table: every rank must lower the same plan, so the
    statistics plans rely on (overflow-free SUM, dense integer GROUP BY) are the all-gathered min / max of
    the shards' values — what the reference's column descriptor holds for all chunks
    (llkv-column-map/src/store/descriptor.rs:19-84)."""
    if world <= 1:
        return
    local = {int(f): table.local_column_stats(int(f)) for f in field_ids}
    gathered = [None] * world
    dist.all_gather_object(gathered, local)
    for f in local:
        pairs = [g[f] for g in gathered if g.get(f) is not None]
        if len(pairs) == world:  # a rank without rows reports its (empty) range too; all must have statistics
            table.set_column_stats(f, min(p[0] for p in pairs), max(p[1] for p in pairs))


def join_groupby_topk(dist, rt, join_agg, rank: int, world: int, limit: int, all_reduce_counts):
    """Drives the collectives of the sharded join → GROUP BY → top-k pipeline (runtime.JoinAgg; dims replicated,
    fact sharded).  ``all_reduce_counts(ptr, n)`` sums the int64 device buffer across ranks in place (RCCL through
    torch on the GPU box; the tests substitute their own).  Returns (rows, total_groups), identical on every rank."""
    ptr, n = join_agg.counts_buffer()
    if world > 1 and n:
        all_reduce_counts(ptr, n)
    g, v = join_agg.straddlers()
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (g, v))
    else:
        gathered = [(g, v)]
    folded = rt.fold_straddlers([x[0] for x in gathered], [x[1] for x in gathered])
    rows, reported = join_agg.candidates(folded, rank, limit)
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (rows, reported))
    else:
        parts = [(rows, reported)]
    merged = rt.merge_join_rows([r for p in parts for r in p[0]], join_agg.n_payload, limit)
    return merged, sum(p[1] for p in parts)


def sorted_groupby(dist, prepared, world: int):
    """GROUP BY of any cardinality over a sharded table (the sort-based route): this rank's shard is reduced on its
    GPU, the partial groups of all ranks are all-gathered and merged in rank order on every rank
    (llkv_hip_query_partial_groups / llkv_hip_query_merge_groups).  Returns the table-wide rows."""
    prepared.launch(0)
    prepared.finish_only()
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, prepared.partial_groups())
        prepared.merge_groups(gathered)
    return prepared.rows()


def distinct_aggregates(dist, prepared, distinct_aggs: Sequence[int], world: int):
    """Ungrouped aggregates with DISTINCT ones among them over a sharded table: the usual exchange-image combine is
    the caller's (launch / all-reduce / finish); this adds the per-aggregate exchange of the ranks' distinct values
    and their merge in rank order.  Returns the rows."""
    if world > 1:
        for a in distinct_aggs:
            gathered = [None] * world
            dist.all_gather_object(gathered, prepared.distinct_partial(a))
            prepared.merge_distinct(a, gathered)
    return prepared.rows()
