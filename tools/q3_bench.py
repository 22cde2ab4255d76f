#!/usr/bin/env python3
"""TPC-H Q3 shape at SF10 on one GPU (BASELINE.json configs[4], single-GPU form): time of
llkv_hip_join_groupby_topk with all inputs resident in HBM; algorithmic bytes per SURVEY.md §8(d)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
D = tpch.DATE_1995_03_15
li = tpch.gen_lineitem(rows, scale, ["l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"])
n_ord = tpch.orders_for_lineitems(rows); od = tpch.gen_orders(n_ord, scale)
n_cust = tpch.customers_for_scale(scale); cu = tpch.gen_customer(n_cust, scale)
lt = rt.HipTable(1, tpch.chunk_rows(rows))
for c in li: lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
ot = rt.HipTable(2, tpch.chunk_rows(n_ord))
for c, (fid, dt) in tpch.ORDERS_SCHEMA.items(): ot.append_column(fid, dt, od[c])
ct = rt.HipTable(3, tpch.chunk_rows(n_cust))
ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
F, O, col = abi.Filter, abi.Operator, abi.col
rev = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))
call = rt.JoinTopk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, rev,
                   payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10, dim_fk=tpch.O_CUSTKEY, dim2=ct,
                   dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
run = call.run  # (the call's C structures are built once, as a C caller holds them)
out, total = run()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); out, total = run(); ts.append(time.perf_counter() - t0)
alg = rows * 28 + n_ord * 28 + n_cust * 9
best = min(ts)
res = {"workload": f"q3_{sf}", "lineitem_rows": rows, "orders": n_ord, "customers": n_cust, "groups": total, "seconds_best": best, "seconds_all": ts,
       "rows_per_s": rows / best, "algorithmic_bytes": alg, "gbs": alg / best / 1e9, "top": out[:3]}
# ---- the general join → GROUP BY route over the same star (llkv_hip_join_groupby_prepare: any aggregate list, any ORDER BY): Q3's own
# aggregate (so the answers can be compared), then a list the hand-tuned pipeline does not take
if "--general" in sys.argv:
    A = abi.AggregateSpec
    lists = {"q3_sum_only": [A.sum(rev)], "two_sums_avg_min_count": [A.sum(rev), A.sum(tpch.L_EXTENDEDPRICE), A.avg(tpch.L_DISCOUNT), A.min(tpch.L_EXTENDEDPRICE), A.count_star()]}
    res["general"] = {}
    for name, aggs in lists.items():
        def once(prepared=None):
            jq = prepared or rt.JoinGroupBy(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, aggs,
                                            dim_fk=tpch.O_CUSTKEY, dim2=ct, dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
            jq.launch(); jq.finish_only()
            rows_, total_ = jq.result([tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], [(abi.JOIN_ORDER_AGGREGATE, 0, True), (abi.JOIN_ORDER_PAYLOAD, 0, False)], 10)
            return jq, rows_, total_
        jq, rows_, total_ = once()
        cold, warm = [], []
        for _ in range(3):
            t0 = time.perf_counter(); j2, rows_, total_ = once(); cold.append(time.perf_counter() - t0); j2.close()
        for _ in range(3):  # the prepared form: the dimension key set is kept, an execution is the fact-side GROUP BY + ORDER BY / LIMIT
            t0 = time.perf_counter(); once(jq); warm.append(time.perf_counter() - t0)
        res["general"][name] = {"groups": total_, "seconds_prepare_and_run_best": min(cold), "seconds_prepared_run_best": min(warm),
                                "top_keys": [r.key for r in rows_[:3]], "same_top_keys_as_the_q3_pipeline": [r.key for r in rows_] == [o[0] for o in out]}
        jq.close()
# ---- configs[4] sharded over `world` ranks, emulated on this one device: what ONE rank runs per query (prepare: its dimension
# work + the probe of its 1/world of lineitem + the run sums; then its share of the exchange) in the general form (dimension
# selection replicated, per-group counts all-reduced) and in the range form (orders of the rank's own key range only, boundary
# runs exchanged).  Not a scaling measurement — one device runs the ranks one after another — but the per-rank time bounds it.
if "--sharded" in sys.argv:
    import torch
    world = 8
    chunks = tpch.chunk_rows(rows)
    kw = lambda t: dict(fact=t, fact_filters=[F(tpch.L_SHIPDATE, O.GreaterThan(D))], fact_key=tpch.L_ORDERKEY, dim=ot, dim_filters=[F(tpch.O_ORDERDATE, O.LessThan(D))],
                        dim_key=tpch.O_ORDERKEY, sum_expr=rev, payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], dim_fk=tpch.O_CUSTKEY, dim2=ct,
                        dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    shards = []
    for r in range(world):
        t = rt.HipTable(1, chunks, r, world)
        lo = sum(chunks[:t.first_chunk])
        for c in li: t.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c][lo:lo + t.local_rows])
        shards.append(t)
    sh = {}
    for form in ("general", "range"):
        per_rank = []
        for r, t in enumerate(shards):
            best_r = 1e9
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                j = rt.JoinAgg(ranged=form == "range", **kw(t))
                if form == "range": blk = j.boundary()
                torch.cuda.synchronize(); best_r = min(best_r, time.perf_counter() - t0)
                del j
            per_rank.append(best_r)
        # the exchange and the finish, once, for the result and the byte count
        joins = [rt.JoinAgg(ranged=form == "range", **kw(t)) for t in shards]
        t0 = time.perf_counter()
        if form == "range":
            blocks = [j.boundary() for j in joins]
            parts = [j.finish_ranged(blocks, r, 10) for r, j in enumerate(joins)]
            exchanged = sum(len(b) for b in blocks)
        else:
            bufs = [j.counts_buffer() for j in joins]
            class Raw: pass
            tens = []
            for p, k in bufs:
                raw = Raw(); raw.__cuda_array_interface__ = {"shape": (int(k),), "typestr": "<i8", "data": (int(p), False), "version": 2}
                tens.append(torch.as_tensor(raw, device="cuda"))
            tot = sum(x.clone() for x in tens)
            for x in tens: x.copy_(tot)
            torch.cuda.synchronize()
            strad = [j.straddlers() for j in joins]
            folded = rt.fold_straddlers([g for g, _ in strad], [v for _, v in strad])
            parts = [j.candidates(folded, r, 10) for r, j in enumerate(joins)]
            exchanged = 2 * 8 * bufs[0][1] * world + sum(12 * len(g) for g, _ in strad)  # all-reduce: every rank's counts out and back
        merged = rt.merge_join_rows([row for rows_r, _ in parts for row in rows_r], 2, 10)
        exchanged += sum(16 + 72 * len(rows_r) for rows_r, _ in parts)
        assert [m[0] for m in merged] == [o[0] for o in out] and [m[1] for m in merged] == [o[1] for o in out], form
        sh[form] = {"per_rank_prepare_ms": [x * 1e3 for x in per_rank], "per_rank_prepare_ms_max": max(per_rank) * 1e3, "exchanged_bytes_per_query": int(exchanged),
                    "single_gpu_ms": best * 1e3, "ratio_to_single_gpu": max(per_rank) / best}
        del joins
    res["sharded_world8_emulated"] = sh
print(json.dumps(res))
