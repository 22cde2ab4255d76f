#!/usr/bin/env python3
"""bench.py — rows/sec + achieved HBM GB/s of the MI355X hot path on TPC-H Q1/Q6.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one execution of the query over the HBM-resident lineitem columns of this rank's
shard: fused scan kernel → octant fold → (N>1: one RCCL all-reduce of the partial aggregate
state) → host finalize.  Default workload: TPC-H Q1 at SF10 (BASELINE.json configs[2]); every
rank holds an SF10 shard (weak scaling: the table is SF(10·N), chunk-sharded), --scaling strong
shards one SF10 table across the ranks instead (configs[3]).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="q1_sf10", help="q1_sf10 | q6_sf10 | q6_sf1 | q1_sf1 | c1_sf0.01 ...")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the workload timed on the CPU oracle (0 = auto)")
    ap.add_argument("--also", default="q6_sf10,q6_sf1,q3_sf10", help="extra workloads measured (N=1 only) and reported under 'also'")
    return ap.parse_args()


def stage(rt, tpch, abi, dist, query, total_rows, scale, rank, world, row_begin_global=0):
    """Generate this rank's shard on the host and stage the needed columns into HBM."""
    dmod = importlib.import_module("rust-llkv_amd.dist")
    chunks = tpch.chunk_rows(total_rows)
    table = rt.HipTable(1, chunks, rank, world)
    first_row = sum(chunks[:table.first_chunk])
    data = tpch.gen_lineitem(table.local_rows, scale, query.columns, row_begin=row_begin_global + first_row)
    t0, (b0, s0) = time.perf_counter(), rt.staging_stats()
    for name in query.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[name]
        if dt == abi.DT_UTF8:
            # one rank: the library finds the dictionary itself; several: the ranks agree on one first
            table.append_utf8_column(fid, data[name], dmod.table_wide_dictionary(dist, data[name], world) if world > 1 else None)
        else:
            table.append_column(fid, dt, data[name])
    # host chunks → pinned ring → hipMemcpyAsync → HBM: the PCIe-bound part.  `copy` = inside the library's copy
    # loop; `wall` adds the binding's side of it (Utf8 dictionary coding, offsets, statistics, Python).
    b1, s1 = rt.staging_stats()
    table.staging = {"wall_seconds": time.perf_counter() - t0, "copy_seconds": s1 - s0, "copy_bytes": b1 - b0}
    dmod.share_column_stats(dist, table, [tpch.LINEITEM_SCHEMA[n][0] for n in query.columns if tpch.LINEITEM_SCHEMA[n][1] != abi.DT_UTF8 and
                                          table.local_column_stats(tpch.LINEITEM_SCHEMA[n][0]) is not None], world)
    return table, data


PROFILE_EVERY = 4
DEPTH = 4  # executions of the prepared query kept in flight (host finalizes i while the GPU runs i+1..)


def run_steps(q, steps, dist, stream_ptr, ex_tensors, torch=None, comm=None):
    """K complete executions; every result is folded and finalized on the host inside the timed region.
    Steady state = one kernel per execution on the compute stream (the scan of execution i folds the tile
    partials of i-1); with several ranks the RCCL all-reduce of an execution's exchange image and its
    copy-out run on a communication stream, one execution behind, beside the next scan."""
    rows, launched, submitted, collected = None, 0, 0, 0

    def exchange(slot):
        if ex_tensors is not None:
            q.wait_folded(comm.cuda_stream)
            with torch.cuda.stream(comm):
                dist.all_reduce(ex_tensors[slot])  # ncclSum over int64 lanes: exact concatenation of shard states
            q.submit(comm.cuda_stream)
        else:
            q.submit(0)

    for i in range(steps):
        if launched - collected == DEPTH:
            if submitted == collected:
                exchange(submitted % DEPTH); submitted += 1
            rows = q.collect(); collected += 1
        q.launch(stream_ptr); launched += 1
        if launched - submitted >= 2:  # the image of the previous execution completed with this launch
            exchange(submitted % DEPTH); submitted += 1
    while collected < launched:
        if submitted == collected:
            exchange(submitted % DEPTH); submitted += 1
        rows = q.collect(); collected += 1
    return rows


def measure(rt, tpch, abi, torch, dist, name, rank, world, scaling, steps, warmup):
    qname, sf = name.split("_")
    query = tpch.QUERIES[qname]()
    rows_sf = tpch.LINEITEM_ROWS[sf]
    scale = tpch.SCALE[sf]
    if scaling == "weak":
        total_rows, gen_scale = rows_sf * world, scale * world
    else:
        total_rows, gen_scale = rows_sf, scale
    if scaling == "weak" and world > 1:
        # every rank holds one SF-sized shard of a (SF·world) table: ragged chunk list, shard r = block r
        chunks = []
        for _ in range(world):
            chunks += tpch.chunk_rows(rows_sf)
        table = rt.HipTable(1, chunks, rank, world)
        first_row = sum(chunks[:table.first_chunk])
        data = tpch.gen_lineitem(table.local_rows, gen_scale, query.columns, row_begin=first_row)
        dmod = importlib.import_module("rust-llkv_amd.dist")
        for cname in query.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[cname]
            if dt == abi.DT_UTF8:
                table.append_utf8_column(fid, data[cname], dmod.table_wide_dictionary(dist, data[cname], world))
            else:
                table.append_column(fid, dt, data[cname])
        dmod.share_column_stats(dist, table, [tpch.LINEITEM_SCHEMA[n][0] for n in query.columns if tpch.LINEITEM_SCHEMA[n][1] != abi.DT_UTF8 and
                                              table.local_column_stats(tpch.LINEITEM_SCHEMA[n][0]) is not None], world)
    else:
        table, data = stage(rt, tpch, abi, dist, query, total_rows, gen_scale, rank, world)
    del data

    q = rt.PreparedQuery(table, query.predicate, query.aggs, query.keys, query.order_by_keys)
    # a dedicated non-default stream shared by the library's kernels and torch's collective,
    # so launch → all-reduce → copy-out are ordered without host synchronisation
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    stream_ptr = stream.cuda_stream
    q.set_depth(DEPTH)
    ex_tensor = None
    if world > 1 or os.environ.get("LLKV_BENCH_FORCE_COLLECTIVE"):  # the env knob rehearses the collective path on one GPU
        # zero-copy int64 views of the library's exchange ring for torch.distributed (RCCL)
        ptr, n = q.exchange_buffer()
        ex_tensor = [_tensor_from_ptr(torch, ptr + slot * n * 8, n) for slot in range(DEPTH)]

    comm = torch.cuda.Stream() if ex_tensor is not None else None
    run_steps(q, warmup, dist, stream_ptr, ex_tensor, torch, comm)
    # HIP events around every 4th scan kernel of the timed region (every one when the region is only a few steps long)
    q.set_profiling(PROFILE_EVERY if steps >= 4 * PROFILE_EVERY else 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = run_steps(q, steps, dist, stream_ptr, ex_tensor, torch, comm)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms, launches, kname = q.kernel_time()
    q.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {
        "name": name, "query": query, "table": table, "prepared": q, "rows_result": rows, "staging": getattr(table, "staging", None),
        "seconds": dt, "total_rows": total_rows, "local_rows": table.local_rows,
        "kernel_ms_avg": kern_ms / max(1, launches), "kernel_launches": launches, "kernel_name": kname,
        "alg_bytes_local": q.algorithmic_bytes, "signature": q.kernel_signature,
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
                return doc[workload]["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def _tensor_from_ptr(torch, ptr, n_i64):
    """int64 CUDA tensor aliasing a raw device pointer (exchange buffer), via __cuda_array_interface__."""

    class _Raw:
        pass

    raw = _Raw()
    raw.__cuda_array_interface__ = {"shape": (int(n_i64),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(raw, device="cuda")


def host_threads():
    """Threads of the parallel CPU baseline: the cores this process may use, at most the 16 a one-GPU box shares out."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(tpch, abi, query, sf, sample_rows):
    """The oracle ("port") timed on this box's host cores on a bounded sample of the workload."""
    from oracle import oracle as orc

    rows = min(sample_rows, tpch.LINEITEM_ROWS[sf])
    data = tpch.gen_lineitem(rows, tpch.SCALE[sf], query.columns)
    t = orc.OracleTable(rows)
    for name in query.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[name]
        t.add(fid, dt, data[name])
    t0 = time.perf_counter()
    if query.grouped:
        orc.groupby(t, query.predicate, query.keys, query.aggs, query.order_by_keys)
    else:
        orc.aggregate(t, query.predicate, query.aggs)
    dt = time.perf_counter() - t0
    out = {"value": rows / dt, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"first {rows} lineitem rows of {query.name}_{sf}, reference-faithful sequential oracle, {dt:.2f} s"}
    try:  # context for both figures: what the host is
        model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
    except OSError:
        model = "unknown"
    out["host"] = {"cpu_model": model, "logical_cpus": os.cpu_count(), "threads_used_parallel": host_threads(),
                   "stream_triad_gbs": round(orc.stream_triad(50_000_000, host_threads()), 1)}
    if query.grouped:
        threads = host_threads()
        try:
            pdt = float("inf")
            for _ in range(3):  # best of three: the first pass also faults the sample's pages into this process
                t0 = time.perf_counter()
                orc.groupby_parallel(t, query.predicate, query.keys, query.aggs, threads)
                pdt = min(pdt, time.perf_counter() - t0)
            out["parallel"] = {"value": rows / pdt, "unit": "rows/s", "cores": threads,
                               "sample": f"{rows} rows, chunk-parallel fused oracle, {pdt:.3f} s"}
        except Exception:  # key shapes the fused mode does not take: only the faithful mode is reported
            pass
    if not query.grouped:
        threads = host_threads()
        prows = min(tpch.LINEITEM_ROWS[sf], max(rows, 20_000_000))
        if prows != rows:
            data = tpch.gen_lineitem(prows, tpch.SCALE[sf], query.columns)
            t = orc.OracleTable(prows)
            for name in query.columns:
                fid, dtp = tpch.LINEITEM_SCHEMA[name]
                t.add(fid, dtp, data[name])
        pdt = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            orc.aggregate_parallel(t, query.predicate, query.aggs, threads)
            pdt = min(pdt, time.perf_counter() - t0)
        out["parallel"] = {"value": prows / pdt, "unit": "rows/s", "cores": threads,
                           "sample": f"{prows} rows, chunk-parallel fused oracle, {pdt:.3f} s"}
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

    def run():
        return rt.join_groupby_topk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, rev,
                                    payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10, dim_fk=tpch.O_CUSTKEY, dim2=ct,
                                    dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    run()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        _, groups = run()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    alg = rows * 28 + n_ord * 28 + n_cust * 9
    for t in (lt, ot, ct):
        t.close()
    return {"rows_per_s": rows / med, "ms_per_step": med * 1e3, "groups": int(groups), "achieved_gbs": alg / med / 1e9, "frac": alg / med / 1e9 / HBM_PEAK_GBS,
            "note": "whole pipeline (2 selections, semi join, hash build, probe, sort, sums, top-k), host-timed median of 7"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    import torch.distributed as dist

    abi = importlib.import_module("rust-llkv_amd.abi")
    rt = importlib.import_module("rust-llkv_amd.runtime")
    tpch = importlib.import_module("rust-llkv_amd.tpch")

    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("LLKV_BENCH_FORCE_COLLECTIVE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    rt.init(local_rank)

    main_res = measure(rt, tpch, abi, torch, dist, args.workload, rank, world, args.scaling, args.steps, args.warmup)
    rows_total = main_res["total_rows"]
    value = rows_total * args.steps / main_res["seconds"]
    kern_s = main_res["kernel_ms_avg"] / 1e3
    achieved = main_res["alg_bytes_local"] / kern_s / 1e9 if kern_s > 0 else 0.0
    qname, sf = args.workload.split("_")
    out = {
        "metric": "rows/sec + HBM GB/s, TPC-H Q1/Q6 SF10 at 1/2/4/8 MI355X",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_res["seconds"] / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"TPC-H {qname.upper()} {sf.upper()} lineitem per GPU ({main_res['local_rows']} rows/GPU, "
                        f"{rows_total} rows total), columns resident in HBM, {main_res['query'].bytes_per_row} B/row algorithmic",
            "sharding": "by chunk (131072 rows) into 8 canonical octants; RCCL all-reduce of partial aggregate state" if world > 1 else "single GPU",
            "kernel": main_res["signature"],
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": (pmc_traffic(args.workload)[0] if world == 1 else None), "traffic_source": pmc_traffic(args.workload)[1],
            "kernel": main_res["kernel_name"], "kernel_ms": main_res["kernel_ms_avg"],
            "algorithmic_bytes_per_launch": main_res["alg_bytes_local"],
        },
        "hbm_gbs_end_to_end": main_res["query"].bytes_per_row * value / 1e9,
    }
    if main_res.get("staging"):
        # never part of `value`: what one cold execution costs when the columns still have to cross PCIe
        st = dict(main_res["staging"])
        st["host_to_hbm_gbs"] = st["copy_bytes"] / max(st["copy_seconds"], 1e-9) / 1e9
        st["rows_per_s_including_staging"] = main_res["total_rows"] / world / (st["wall_seconds"] + main_res["seconds"] / args.steps)
        out["staging"] = st

    if rank == 0 and world == 1:
        also = {}
        # free the main workload's HBM image before staging the next one
        main_q = main_res.pop("prepared"); main_q.close()
        main_res.pop("table").close()
        for name in [w for w in args.also.split(",") if w and w != args.workload]:
            if name.startswith("q3_"):
                also[name] = measure_q3(rt, tpch, abi, name.split("_")[1])
                continue
            r = measure(rt, tpch, abi, torch, dist, name, 0, 1, "weak", args.steps, args.warmup)
            ks = r["kernel_ms_avg"] / 1e3
            also[name] = {"rows_per_s": r["total_rows"] * args.steps / r["seconds"], "ms_per_step": r["seconds"] / args.steps * 1e3,
                          "kernel_ms": r["kernel_ms_avg"], "achieved_gbs": r["alg_bytes_local"] / ks / 1e9 if ks else 0.0,
                          "frac": (r["alg_bytes_local"] / ks / 1e9 / HBM_PEAK_GBS) if ks else 0.0}
            r["prepared"].close(); r["table"].close()
        out["also"] = also
        if not args.no_cpu_baseline:
            sample = args.cpu_sample_rows or (40_000_000 if main_res["query"].grouped else 60_000_000)  # ≈10–20 s of CPU work
            out["cpu_baseline"] = cpu_baseline(tpch, abi, main_res["query"], sf, sample)
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
