#!/usr/bin/env python3
"""bench.py — rows/sec + achieved HBM GB/s of the MI355X hot path on TPC-H Q1/Q6.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one execution of the query over the HBM-resident lineitem columns of this rank's
shard: fused scan kernel → octant fold → (N>1: one RCCL all-reduce of the partial aggregate
state) → host finalize.  Default workload: TPC-H Q1 at SF10 (BASELINE.json configs[2]); with
N > 1 the ONE SF10 table is sharded by chunk across the ranks (strong scaling, configs[3] — the
configuration `metric` is quoted on); --scaling weak gives every rank an SF10 shard of an
SF(10·N) table instead, and is also reported under `also.q1_sf10_weak`.

Started as plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) with N > 1, this
process only launches `python -m torch.distributed.run --nproc-per-node N bench.py …` as a CHILD
(before anything touches the GPU), relays its output and exits with its code.
`--dry-run-layout` stops before device binding: the ranks meet over gloo, build their shard of the
workload's chunk list through the library's host code and rank 0 prints the layout (CPU rehearsal).
"""
from __future__ import annotations

import argparse
import re
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="q1_sf10", help="q1_sf10 | q6_sf10 | q6_sf1 | q1_sf1 | c1_sf0.01 ...")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1: strong = one SF table sharded over the ranks (BASELINE configs[3]); weak = one SF shard per rank")
    ap.add_argument("--dry-run-layout", action="store_true", help="no GPU: gloo rendezvous, shard layout per rank, exit")
    ap.add_argument("--dry-run-comm-failure", type=int, default=-1, help="with --dry-run-layout: this rank pretends it could not join the communicator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the workload timed on the CPU oracle (0 = auto)")
    ap.add_argument("--also", default="q6_sf10,q6_sf1,q3_sf10,q1_sf10_decimal", help="extra workloads reported under 'also' (N>1: q1 weak, q6_sf10 and q3_sf10 sharded)")
    return ap.parse_args()


def stage(rt, tpch, abi, dist, query, chunks, scale, rank, world, row_begin_global=0, twice=False, decimal=False):
    """Generate this rank's shard (its chunks of the global chunk list) on the host and stage the needed columns into HBM.
    `decimal`: quantity / price / discount / tax as DECIMAL(15,2) — the reference's own TPC-H DDL — handed over as arrow's
    16-byte Decimal128 raw values."""
    table = rt.HipTable(1, chunks, rank, world)
    first_row = sum(chunks[:table.first_chunk])
    data = tpch.gen_lineitem(table.local_rows, scale, query.columns, row_begin=row_begin_global + first_row)
    if decimal:
        data = tpch.lineitem_as_decimal(data)
    t0, (b0, s0) = time.perf_counter(), rt.staging_stats()
    for name in query.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[name][0], tpch.lineitem_dtype(name, decimal)
        if dt == abi.DT_DECIMAL128:
            table.append_decimal128_column(fid, tpch.DECIMAL_PRECISION, tpch.DECIMAL_SCALE, data[name])
        elif dt == abi.DT_UTF8:
            # one rank: the library finds the dictionary itself; several: the ranks agree on one first (sorted union of
            # the shards' distinct values, all-gathered over the library's communicator)
            table.append_utf8_column(fid, data[name], rt.comm_union_strings(sorted(chr(int(v)) for v in np.unique(data[name]))) if world > 1 else None)
        else:
            table.append_column(fid, dt, data[name])
    # host chunks → pinned ring → hipMemcpyAsync → HBM: the PCIe-bound part.  `copy` = inside the library's copy
    # loop; `wall` adds the binding's side of it (Utf8 dictionary coding, offsets, statistics, Python).
    b1, s1 = rt.staging_stats()
    table.staging = {"wall_seconds": time.perf_counter() - t0, "copy_seconds": s1 - s0, "copy_bytes": b1 - b0}
    if world == 1 and twice:
        # the same columns into a second table: what staging costs once the process has its copy lanes, its pinned blocks and its
        # first device allocations behind it (the first table of a process pays for those too)
        t0, (b0, s0) = time.perf_counter(), rt.staging_stats()
        again = rt.HipTable(2, chunks, rank, world)
        for name in query.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[name][0], tpch.lineitem_dtype(name, decimal)
            if dt == abi.DT_DECIMAL128:
                again.append_decimal128_column(fid, tpch.DECIMAL_PRECISION, tpch.DECIMAL_SCALE, data[name])
            else:
                again.append_utf8_column(fid, data[name]) if dt == abi.DT_UTF8 else again.append_column(fid, dt, data[name])
        b1, s1 = rt.staging_stats()
        table.staging["second_table"] = {"wall_seconds": time.perf_counter() - t0, "copy_seconds": s1 - s0, "copy_bytes": b1 - b0,
                                         "host_to_hbm_gbs": (b1 - b0) / max(s1 - s0, 1e-9) / 1e9}
        again.close()
    table.share_metadata()  # sharded: table-wide integer statistics and NULL-ability, so every rank lowers the same plan
    return table, data


PROFILE_EVERY = 8  # HIP events bracket every 8th scan of the timed region: each record is a ~4 µs packet between back-to-back kernels
DEPTH = 4  # executions of the prepared query kept in flight (host finalizes i while the GPU runs i+1..)


def run_steps(q, steps, stream_ptr, comm_stream_ptr, comm_stream=None, allreduce_events=None):
    """K complete executions; every result is folded and finalized on the host inside the timed region.
    Steady state = one kernel per execution on the compute stream (the scan of execution i folds the tile
    partials of i-1); with several ranks the RCCL all-reduce of an execution's exchange image (inside the library:
    llkv_hip_query_all_reduce → ncclAllReduce, int64 sum = exact concatenation of the shard states) and its copy-out
    run on a communication stream, one execution behind, beside the next scan."""
    rows, launched, submitted, collected = None, 0, 0, 0

    def exchange(slot):
        if comm_stream_ptr:
            # every 4th all-reduce of a timed region is bracketed by events on the communication stream (`collective.allreduce_us`)
            timed = allreduce_events is not None and comm_stream is not None and (submitted % PROFILE_EVERY) == 0
            if timed:
                import torch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(comm_stream)
            q.all_reduce(comm_stream_ptr)
            if timed:
                e1.record(comm_stream)
                allreduce_events.append((e0, e1))
            q.submit(comm_stream_ptr)
        else:
            q.submit(0)

    # collect_only: the library waits, folds the octants and finalizes every group of the execution (that IS the result); the
    # binding's Python row objects are built once, from the last execution — 40 ctypes calls per execution are the binding's
    # cost, not the path's, and behind the last launch nothing hides them
    for i in range(steps):
        if launched - collected == DEPTH:
            if submitted == collected:
                exchange(submitted % DEPTH); submitted += 1
            q.collect_only(); collected += 1
        q.launch(stream_ptr); launched += 1
        if launched - submitted >= 2:  # the image of the previous execution completed with this launch
            exchange(submitted % DEPTH); submitted += 1
    while collected < launched:
        if submitted == collected:
            exchange(submitted % DEPTH); submitted += 1
        q.collect_only(); collected += 1
    return q.rows() if launched else None


def split_workload(name):
    """'q1_sf10' → ('q1', 'sf10', False); 'q1_sf10_decimal' → ('q1', 'sf10', True): the same rows with DECIMAL(15,2) money columns."""
    parts = name.split("_")
    return parts[0], parts[1], len(parts) > 2 and parts[2] == "decimal"


def measure(rt, tpch, abi, torch, dist, name, rank, world, scaling, steps, warmup, stage_twice=False):
    qname, sf, decimal = split_workload(name)
    query = tpch.QUERIES[qname]()
    chunks, total_rows, gen_scale = workload_chunks(tpch, name, scaling, world)
    table, data = stage(rt, tpch, abi, dist, query, chunks, gen_scale, rank, world, twice=stage_twice, decimal=decimal)
    del data

    q = rt.PreparedQuery(table, query.predicate, query.aggs, query.keys, query.order_by_keys)
    # a dedicated non-default stream shared by the library's kernels and torch's collective,
    # so launch → all-reduce → copy-out are ordered without host synchronisation
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    stream_ptr = stream.cuda_stream
    q.set_depth(DEPTH)
    # the env knob rehearses the collective path on one GPU (a one-rank RCCL communicator)
    comm = torch.cuda.Stream() if (world > 1 or os.environ.get("LLKV_BENCH_FORCE_COLLECTIVE")) else None
    comm_ptr = comm.cuda_stream if comm is not None else 0
    run_steps(q, warmup, stream_ptr, comm_ptr)
    ar_events = [] if comm is not None else None
    # HIP events around every 8th scan kernel of the timed region (every 4th / every one when the region is only a few steps long)
    q.set_profiling(PROFILE_EVERY if steps >= 2 * PROFILE_EVERY else (4 if steps >= 8 else 1))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = run_steps(q, steps, stream_ptr, comm_ptr, comm, ar_events)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms, launches, kname = q.kernel_time()
    q.set_profiling(False)
    collective = None
    if comm is not None:
        backend, ranks = rt.comm_describe()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ar_events)
        collective = {"backend": backend, "ranks": ranks, "bytes_per_step": int(q.exchange_buffer()[1]) * 8,
                      "allreduce_us": us[len(us) // 2] if us else None, "allreduce_samples": len(us),
                      "what": "one all-reduce (int64 sum) of the exchange image [8 octants][lanes] per execution, on the communication stream, "
                              "one execution behind the scan; median of event-bracketed samples on rank 0"}
    kernel_ms_by_rank = [kern_ms / max(1, launches)]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, kernel_ms_by_rank[0])
        kernel_ms_by_rank = gathered
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {
        "name": name, "query": query, "table": table, "prepared": q, "rows_result": rows, "staging": getattr(table, "staging", None),
        "seconds": dt, "total_rows": total_rows, "local_rows": table.local_rows,
        "kernel_ms_avg": kern_ms / max(1, launches), "kernel_launches": launches, "kernel_name": kname,
        "alg_bytes_local": q.algorithmic_bytes, "signature": q.kernel_signature,
        "collective": collective, "kernel_ms_by_rank": kernel_ms_by_rank,
    }
    return res


def pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN/pmc_traffic.json: 2·FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md §HBM).  Counters cannot be collected from inside this process."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            if workload in doc:
                meta = doc.get("_meta", {})
                src = os.path.relpath(path, ROOT) + (f" (collected {meta.get('date')}, commit {meta.get('commit')})" if meta else "")
                return doc[workload]["traffic_bytes_per_launch"], src
        except Exception:
            continue
    return None, None


def pmc_traffic_q3():
    """HBM bytes of one Q3 query, summed over its kernels, from the committed PMC pass (profiles/rNN/pmc_q3.json)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_q3.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            return sum(v["traffic_bytes"] for v in doc.values() if isinstance(v, dict) and "traffic_bytes" in v), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def with_bus_fraction(entry, traffic, src, seconds):
    """`frac` is ALGORITHMIC bytes over time; where a kernel reads fewer bytes than that (late materialisation) the bus
    sees less: `traffic_bytes` = the committed PMC figure, `bus_gbs` / `bus_frac` = that over the same time against 8 TB/s."""
    if traffic and seconds > 0:
        entry["traffic_bytes"] = traffic
        entry["traffic_source"] = src
        entry["bus_gbs"] = traffic / seconds / 1e9
        entry["bus_frac"] = traffic / seconds / 1e9 / HBM_PEAK_GBS
    return entry


def detected_threads():
    """std::thread::available_parallelism as the reference's pool sees it (llkv-threading/src/lib.rs:15-20): the CPUs
    this process may run on, capped by a cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, -(-int(parts[0]) // int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, -(-quota // period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def host_threads():
    """Threads of the parallel CPU baseline — `configured_thread_count` of the reference (llkv-threading/src/lib.rs:22-31):
    LLKV_MAX_THREADS when it parses to a positive number, else every core this process may use."""
    try:
        v = int(os.environ.get("LLKV_MAX_THREADS", "").strip())
        if v > 0:
            return v
    except ValueError:
        pass
    return detected_threads()


def cpu_baseline(tpch, abi, query, sf, sample_rows, decimal=False):
    """The oracle ("port") timed on this box's host cores.  Faithful leg (the reference's pass structure, one core for
    Q1/Q6 exactly like the reference): a bounded prefix of the workload; parallel leg (chunk-parallel fused mode over
    LLKV_MAX_THREADS / all usable cores): the FULL workload."""
    from oracle import oracle as orc

    full = tpch.LINEITEM_ROWS[sf]
    data = tpch.gen_lineitem(full, tpch.SCALE[sf], query.columns)
    if decimal:
        data = tpch.lineitem_as_decimal(data)

    def table_of(n):
        t = orc.OracleTable(n)
        for name in query.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[name][0], tpch.lineitem_dtype(name, decimal)
            if dt == abi.DT_DECIMAL128:
                t.add(fid, dt, data[name][:n], precision=tpch.DECIMAL_PRECISION, scale=tpch.DECIMAL_SCALE)
            else:
                t.add(fid, dt, data[name][:n])
        return t
    whole = table_of(full)
    rows = min(sample_rows, full)
    t = whole if rows == full else table_of(rows)
    t0 = time.perf_counter()
    if query.grouped:
        orc.groupby(t, query.predicate, query.keys, query.aggs, query.order_by_keys)
    else:
        orc.aggregate(t, query.predicate, query.aggs)
    dt = time.perf_counter() - t0
    out = {"value": rows / dt, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"first {rows} of {full} lineitem rows of {query.name}_{sf}, reference-faithful sequential oracle, {dt:.2f} s"}
    try:  # context for both figures: what the host is
        model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    threads = host_threads()
    out["host"] = {"cpu_model": model, "logical_cpus": os.cpu_count(), "usable_cpus": detected_threads(),
                   "LLKV_MAX_THREADS": os.environ.get("LLKV_MAX_THREADS"), "threads_used_parallel": threads,
                   "stream_triad_gbs": round(orc.stream_triad(50_000_000, threads), 1)}
    try:
        pdt = float("inf")
        for _ in range(3):  # best of three: the first pass also faults the pages into this process
            t0 = time.perf_counter()
            if query.grouped:
                orc.groupby_parallel(whole, query.predicate, query.keys, query.aggs, threads)
            else:
                orc.aggregate_parallel(whole, query.predicate, query.aggs, threads)
            pdt = min(pdt, time.perf_counter() - t0)
        out["parallel"] = {"value": full / pdt, "unit": "rows/s", "cores": threads,
                           "sample": f"all {full} rows of {query.name}_{sf}, chunk-parallel fused oracle, best of 3: {pdt:.3f} s"}
    except Exception:  # key shapes the fused mode does not take: only the faithful mode is reported
        pass
    return out


def measure_q3(rt, tpch, abi, sf):
    """BASELINE.json configs[4], single-GPU form: customer(segment) ⋉ orders(date) ⋈ lineitem(shipdate), GROUP BY the
    order, SUM(price·(1−disc)), top 10 — wall time of llkv_hip_join_groupby_topk with every input resident in HBM
    (hash probing is latency bound: GB/s are quoted against the algorithmic bytes of SURVEY.md §8d for reference)."""
    rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
    D = tpch.DATE_1995_03_15
    li = tpch.gen_lineitem(rows, scale, ["l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"])
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    lt = rt.HipTable(1, tpch.chunk_rows(rows))
    for c in li:
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
    ot = rt.HipTable(2, tpch.chunk_rows(n_ord))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    del li, od, cu
    F, O, col = abi.Filter, abi.Operator, abi.col
    rev = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))

    # (the C structures of the call are built once, as a C caller holds them: the timed region is llkv_hip_join_groupby_topk itself)
    call = rt.JoinTopk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, rev,
                       payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10, dim_fk=tpch.O_CUSTKEY, dim2=ct,
                       dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    run = call.run
    run()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        _, groups = run()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    alg = rows * 28 + n_ord * 28 + n_cust * 9
    for t in (lt, ot, ct):
        t.close()
    return {"rows_per_s": rows / med, "ms_per_step": med * 1e3, "groups": int(groups), "achieved_gbs": alg / med / 1e9, "frac": alg / med / 1e9 / HBM_PEAK_GBS,
            "note": "whole pipeline (fill, two key-set scans, probe with the bitmap ranks on the way, run sums + slice winners, top-k: 6 launches), host-timed median of 9; "
                    "the probe streams 8 B of every lineitem row (ship date + the 4-byte image of the order key) and reads price / discount for the rows that join "
                    "(late materialisation), the order scan 12 B of every order: "
                    "GB/s and frac are ALGORITHMIC bytes (28 B per row) over time, the HBM traffic is below them"}


def measure_q3_sharded(rt, tpch, abi, torch, dist, sf, rank, world):
    """BASELINE.json configs[4]: build side (orders, customer) replicated on every rank, lineitem sharded by chunk; per
    step = local build + probe + per-group sums (JoinAgg) then llkv_hip_join_agg_finish_sharded.  The range form first
    (lineitem is clustered by the order key: every rank selects the orders of its own key range only and the ranks
    exchange ~1 KB of boundary runs + their candidates); if a rank refuses it, the general form (one int64 all-reduce of
    the per-group row counts over RCCL, all-gathers of the straddlers and of <= limit candidates)."""
    rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
    D = tpch.DATE_1995_03_15
    chunks = tpch.chunk_rows(rows)
    lt = rt.HipTable(1, chunks, rank, world)
    first_row = sum(chunks[:lt.first_chunk])
    cols = ["l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"]
    li = tpch.gen_lineitem(lt.local_rows, scale, cols, row_begin=first_row)
    for c in cols:
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
    lt.share_metadata()
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    ot = rt.HipTable(2, tpch.chunk_rows(n_ord))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    del li, od, cu
    F, O, col = abi.Filter, abi.Operator, abi.col
    rev = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))

    state = {"ranged": True, "bytes": 0}

    def run():
        def once(ranged):
            ja = rt.JoinAgg(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, rev,
                            payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], dim_fk=tpch.O_CUSTKEY, dim2=ct,
                            dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY, ranged=ranged)
            out = ja.finish_sharded(10)  # the collectives run inside the library (RCCL)
            state["bytes"] = ja.exchange_bytes()
            return out
        if state["ranged"]:
            # prepare refuses a shape that does not qualify on this rank, finish_sharded a pair stream out of key order
            # (on every rank alike): all ranks must take the same form, so they agree first
            ok = 1
            res = None
            try:
                ja_try = rt.JoinAgg(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, rev,
                                    payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], dim_fk=tpch.O_CUSTKEY, dim2=ct,
                                    dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY, ranged=True)
            except abi.LlkvError:
                ok, ja_try = 0, None
            flag = torch.tensor([ok], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                try:
                    res = ja_try.finish_sharded(10)
                    state["bytes"] = ja_try.exchange_bytes()
                    return res
                except abi.LlkvError:
                    pass
            state["ranged"] = False
        return once(False)

    run()
    ts = []
    for _ in range(7):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, groups = run()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        ts.append(float(dt.item()))
    ts.sort()
    med = ts[len(ts) // 2]
    alg = rows * 28 + world * (n_ord * 28 + n_cust * 9)  # every rank reads the replicated build side
    for t in (lt, ot, ct):
        t.close()
    return {"rows_per_s": rows / med, "ms_per_step": med * 1e3, "groups": int(groups), "achieved_gbs": alg / med / 1e9,
            "frac": alg / med / 1e9 / (HBM_PEAK_GBS * world), "scaling": "strong", "form": "range" if state["ranged"] else "general",
            "exchanged_bytes_per_query": int(state["bytes"]),
            "note": "probe side sharded by chunk, build side replicated; max over ranks, host-timed median of 7, collectives included"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks with torch.distributed.run as a CHILD process
    (this process never touches the GPU — a process that has initialised it must not exec another program), relay
    the child's output and return its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def require_all_ranks(torch, dist, ok, why, rank, device):
    """A measurement has no fallback transport: every rank learns whether ALL of them joined the library's RCCL communicator, and
    all of them leave with exit code 3 otherwise (LLKV_BENCH_HOST_TRANSPORT=1 is the rehearsal on one device, and says so in its
    line).  `device`: where the agreement tensor lives ("cuda" over nccl; "cpu" in the gloo rehearsal of this very path)."""
    agreed = torch.tensor([1 if ok else 0], device=device)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    if int(agreed.item()) == 0:
        sys.stderr.write(f"bench.py rank {rank}: the library's RCCL communicator could not be created"
                         + (f" ({why})" if why else " on another rank") + " — no measurement without it\n")
        dist.destroy_process_group()
        raise SystemExit(3)


def workload_chunks(tpch, name, scaling, world):
    """(global chunk list, table rows, generator scale) of a lineitem workload under a scaling mode."""
    qname, sf, _ = split_workload(name)
    rows_sf, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
    if scaling == "weak" and world > 1:
        chunks = []
        for _ in range(world):  # every rank holds one SF-sized shard of an (SF·world) table: shard r = block r
            chunks += tpch.chunk_rows(rows_sf)
        return chunks, rows_sf * world, scale * world
    return tpch.chunk_rows(rows_sf), rows_sf, scale


def dry_run_layout(args, rank, world):
    """CPU rehearsal of the launch path: gloo rendezvous, every rank builds its shard of the chunk list through the
    library's host code (llkv_hip_table_create / llkv_hip_shard_layout — no device), rank 0 prints the layout."""
    import torch.distributed as dist
    rt = importlib.import_module("rust-llkv_amd.runtime")
    tpch = importlib.import_module("rust-llkv_amd.tpch")
    dmod = importlib.import_module("rust-llkv_amd.dist")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.dry_run_comm_failure >= 0 and world > 1:
        # rehearsal of the measurement's "no fallback" rule over gloo: the named rank pretends its ncclCommInitRank failed
        import torch
        failed = rank == args.dry_run_comm_failure
        require_all_ranks(torch, dist, not failed, "rehearsed failure" if failed else "", rank, "cpu")
    chunks, total_rows, _ = workload_chunks(tpch, args.workload, args.scaling, world)
    table = rt.HipTable(1, chunks, rank, world)
    begin, owner = dmod.shard_layout(rt.lib(), len(chunks), world)
    mine = {"rank": rank, "first_chunk": table.first_chunk, "n_chunks": table.n_local_chunks, "local_rows": table.local_rows,
            "first_row": sum(chunks[:table.first_chunk]), "octants": [o for o in range(8) if owner[o] == rank]}
    gathered = [mine]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
    if rank == 0:
        print(json.dumps({"dry_run_layout": True, "n_gpus": world, "scaling": args.scaling, "workload": args.workload, "total_rows": total_rows,
                          "n_chunks": len(chunks), "octant_chunk_begin": begin, "ranks": gathered,
                          "rows_covered": sum(g["local_rows"] for g in gathered)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def also_entry(r, steps):
    ks = r["kernel_ms_avg"] / 1e3
    return {"rows_per_s": r["total_rows"] * steps / r["seconds"], "ms_per_step": r["seconds"] / steps * 1e3,
            "kernel_ms": r["kernel_ms_avg"], "achieved_gbs": r["alg_bytes_local"] / ks / 1e9 if ks else 0.0,
            "frac": (r["alg_bytes_local"] / ks / 1e9 / HBM_PEAK_GBS) if ks else 0.0, "local_rows": r["local_rows"]}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if args.dry_run_layout:
        return dry_run_layout(args, rank, world)
    import torch
    import torch.distributed as dist

    abi = importlib.import_module("rust-llkv_amd.abi")
    rt = importlib.import_module("rust-llkv_amd.runtime")
    tpch = importlib.import_module("rust-llkv_amd.tpch")

    # LLKV_BENCH_HOST_TRANSPORT=1: rehearsal of the N-rank orchestration on ONE GPU — every rank binds device 0, the
    # library's collectives go through a host transport over gloo (llkv_hip_comm_init_custom) instead of RCCL, which
    # refuses two ranks on one device.  Not a measurement.
    rehearsal = bool(os.environ.get("LLKV_BENCH_HOST_TRANSPORT"))
    collective_backend = "host transport over gloo (rehearsal)" if rehearsal else "RCCL"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("LLKV_BENCH_FORCE_COLLECTIVE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    rt.init(local_rank)
    if dist.is_initialized() and rehearsal:
        rt.comm_init_torch(dist, rank, world)
    elif dist.is_initialized():
        # the library's own RCCL communicator (the collectives of the data path); torch.distributed only carries the
        # ncclUniqueId, the barriers and the max-over-ranks clock of the bench contract
        uid = [rt.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ok, why = 1, ""
        try:
            rt.comm_init(uid[0], rank, world)
            backend, ranks = rt.comm_describe()
            if backend != "rccl" or ranks != world:
                ok, why = 0, f"the communicator reports backend {backend} with {ranks} ranks, {world} were launched"
        except Exception as e:
            ok, why = 0, str(e)
        require_all_ranks(torch, dist, ok, why, rank, "cuda")

    main_res = measure(rt, tpch, abi, torch, dist, args.workload, rank, world, args.scaling, args.steps, args.warmup, stage_twice=True)
    rows_total = main_res["total_rows"]
    value = rows_total * args.steps / main_res["seconds"]
    kern_s = main_res["kernel_ms_avg"] / 1e3
    achieved = main_res["alg_bytes_local"] / kern_s / 1e9 if kern_s > 0 else 0.0
    qname, sf, main_decimal = split_workload(args.workload)
    traffic, traffic_src = pmc_traffic(args.workload) if world == 1 else (None, None)
    if world == 1:
        shape = f"TPC-H {qname.upper()} {sf.upper()} lineitem on one GPU ({rows_total} rows)"
    elif args.scaling == "strong":
        shape = f"TPC-H {qname.upper()} {sf.upper()} lineitem ({rows_total} rows) sharded by chunk over {world} GPUs ({main_res['local_rows']} rows on rank 0)"
    else:
        shape = f"TPC-H {qname.upper()} lineitem, one {sf.upper()} shard per GPU ({main_res['local_rows']} rows/GPU, {rows_total} rows total)"
    out = {
        "metric": "rows/sec + HBM GB/s, TPC-H Q1/Q6 SF10 at 1/2/4/8 MI355X",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_res["seconds"] / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "i64 (exact decimal)" if main_decimal else "f64", "data": "synthetic",
        "config": {
            "workload": f"{shape}, columns resident in HBM, {main_res['query'].bytes_per_row} B/row algorithmic",
            "sharding": (f"{args.scaling}: by chunk (131072 rows) into 8 canonical octants, rank r owns octants [r·8/N, (r+1)·8/N); "
                         f"one all-reduce (int64 sum) of the partial aggregate state per execution, {collective_backend}") if world > 1 else "single GPU",
            "kernel": main_res["signature"],
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # PMC counters cannot be collected from inside this process: `traffic` is the committed rocprofv3 --pmc result
            # for this workload and kernel (tools/pmc_traffic.sh), NOT a measurement of this run
            "traffic": traffic, "traffic_measured_in_run": False, "traffic_source": traffic_src,
            "kernel": main_res["kernel_name"], "kernel_ms": main_res["kernel_ms_avg"],
            "algorithmic_bytes_per_launch": main_res["alg_bytes_local"], "rank": 0,
        },
        "hbm_gbs_end_to_end": main_res["query"].bytes_per_row * value / 1e9,
    }
    if main_res.get("collective"):
        out["collective"] = main_res["collective"]
    if world > 1:
        out["kernel_ms_by_rank"] = main_res["kernel_ms_by_rank"]
    if main_res.get("staging"):
        # never part of `value`: what one cold execution costs when the columns still have to cross PCIe
        st = dict(main_res["staging"])
        st["host_to_hbm_gbs"] = st["copy_bytes"] / max(st["copy_seconds"], 1e-9) / 1e9
        st["rows_per_s_including_staging"] = main_res["total_rows"] / world / (st["wall_seconds"] + main_res["seconds"] / args.steps)
        out["staging"] = st

    # free the main workload's HBM image before staging the next one
    main_res.pop("prepared").close()
    main_res.pop("table").close()
    also = {}
    names = [w for w in args.also.split(",") if w]
    if world == 1:
        for name in [w for w in names if w != args.workload]:
            if name.startswith("q3_"):
                also[name] = measure_q3(rt, tpch, abi, name.split("_")[1])
                if name == "q3_sf10":
                    t3, src3 = pmc_traffic_q3()
                    with_bus_fraction(also[name], t3, src3, also[name]["ms_per_step"] / 1e3)
                continue
            r = measure(rt, tpch, abi, torch, dist, name, 0, 1, "strong", args.steps, args.warmup)
            also[name] = also_entry(r, args.steps)
            tr, tsrc = pmc_traffic(name)
            with_bus_fraction(also[name], tr, tsrc, r["kernel_ms_avg"] / 1e3)
            if split_workload(name)[2]:
                also[name]["bytes_per_row"] = r["query"].bytes_per_row
                also[name]["dtype"] = "i64 (exact decimal)"
                also[name]["note"] = ("quantity / price / discount / tax as DECIMAL(15,2) (the reference's own TPC-H DDL), staged from arrow's 16-byte "
                                      "Decimal128 values narrowed to 8 B/row: the computed sums are exact integer lanes (PlanValue Decimal arm), "
                                      "every cell bit-equal to the reference's Decimal128(p, s)")
            if tpch.LINEITEM_ROWS[name.split("_")[1]] * r["query"].bytes_per_row < 256 << 20:
                also[name]["note"] = ("the columns fit the 256 MB Infinity Cache and are re-scanned every step: not an HBM figure; "
                                      f"launch-bound: {also[name]['ms_per_step'] * 1e3 - also[name]['kernel_ms'] * 1e3:.1f} µs per step outside the kernel")
            if re.search(r",0,1,\d+>$", r["prepared"].kernel_signature):
                also[name]["note"] = (also[name].get("note", "") + " Argument-only columns are read for the rows that pass (Plan::EARLY): "
                                      "achieved / frac are algorithmic bytes over time, the HBM traffic is below them.").strip()
            r["prepared"].close(); r["table"].close()
    else:
        # every rank takes part (collectives inside); rank 0 reports
        other = "weak" if args.scaling == "strong" else "strong"
        r = measure(rt, tpch, abi, torch, dist, args.workload, rank, world, other, args.steps, args.warmup)
        also[f"{args.workload}_{other}"] = dict(also_entry(r, args.steps), scaling=other, total_rows=r["total_rows"])
        r["prepared"].close(); r["table"].close()
        for name in [w for w in names if w != args.workload and split_workload(w)[1] == sf]:
            if name.startswith("q3_"):
                also[name] = measure_q3_sharded(rt, tpch, abi, torch, dist, name.split("_")[1], rank, world)
            else:
                r = measure(rt, tpch, abi, torch, dist, name, rank, world, args.scaling, args.steps, args.warmup)
                also[name] = dict(also_entry(r, args.steps), scaling=args.scaling)
                r["prepared"].close(); r["table"].close()
    out["also"] = also
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sample = args.cpu_sample_rows or (40_000_000 if main_res["query"].grouped else 60_000_000)  # ≈10–20 s of CPU work
        out["cpu_baseline"] = cpu_baseline(tpch, abi, main_res["query"], sf, sample, main_decimal)
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        rt.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
