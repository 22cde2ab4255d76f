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
        local = sorted({chr(int(v)) for v in np.unique(local_values)})
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
