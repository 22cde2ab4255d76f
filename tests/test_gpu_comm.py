"""GPU: the collectives behind the C ABI (include/llkv_hip.h "Collectives").
  * RCCL transport: a one-rank communicator on cuda:0 — ncclCommInitRank, ncclAllReduce on the exchange image, the
    all-gathers bounced through HBM (what N ranks run over xGMI; the driver's multi-GPU bench exercises N > 1).
  * the sharded drivers end to end on TWO processes that share the one GPU of the test box, over a host-supplied
    transport (gloo behind llkv_hip_comm_init_custom; RCCL refuses two ranks on one device): dense GROUP BY
    (finish_sharded), sort-based GROUP BY, DISTINCT, the join → GROUP BY → top-k pipeline, and the metadata agreement
    that keeps the ranks' plans identical when only one shard has NULL cells.
Every answer is compared with the single-GPU answer of the same library (itself checked against the oracle in
test_gpu_parity.py): integers, keys, order exact; f64 sums bit-exact where the design promises it."""
import dataclasses
import os
import pickle
import socket

import numpy as np
import pytest

from conftest import mod

pytestmark = pytest.mark.gpu
REL = 1e-9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(rows):
    return [(tuple(k.value for k in r.keys), tuple(np.float64(v.value).tobytes() if isinstance(v.value, float) else v.value for v in r.values)) for r in rows]


def test_rccl_one_rank_communicator(rt, abi, tpch):
    import torch
    uid = rt.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    rt.comm_init(uid, 0, 1)
    try:
        assert rt.comm_world() == 1
        with pytest.raises(abi.LlkvError):  # one communicator per process
            rt.comm_init(uid, 0, 1)
        t = torch.arange(-500, 500, dtype=torch.int64, device="cuda")
        want = t.clone()
        rt.comm_all_reduce_i64(t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(t, want)  # a sum over one rank
        assert rt.comm_all_gather_v(b"hello xGMI") == [b"hello xGMI"]
        assert rt.comm_all_gather_v(b"") == [b""]
        assert rt.comm_union_strings(["b", "a", "b"]) == ["a", "b"]
        # the query's collective inside the library: launch → all_reduce → submit → collect
        n = 300_000
        d = tpch.gen_lineitem(n, 1.0)
        q = tpch.q1()
        ht = rt.HipTable(1, tpch.chunk_rows(n, 32768))
        for c in q.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[c]
            ht.append_utf8_column(fid, d[c]) if dt == abi.DT_UTF8 else ht.append_column(fid, dt, d[c])
        ht.share_metadata()  # a no-op on one rank
        pq = rt.PreparedQuery(ht, q.predicate, q.aggs, q.keys, True)
        want_rows = _flat(pq.run())
        pq.set_depth(4)
        stream = torch.cuda.Stream()
        for _ in range(3):
            pq.launch(stream.cuda_stream)
        for _ in range(3):
            pq.all_reduce(stream.cuda_stream)
            pq.submit(stream.cuda_stream)
            assert _flat(pq.collect()) == want_rows
        pq.launch(0)
        assert _flat(pq.finish_sharded()) == want_rows
    finally:
        rt.comm_destroy()
    assert rt.comm_world() == 0


# ---------------------------------------------------------------------------------------------------------------
# two processes, one GPU, host transport
# ---------------------------------------------------------------------------------------------------------------
def _tables(rt, abi, rank, world):
    """Deterministic inputs every rank can regenerate; returns this rank's shards."""
    rng = np.random.default_rng(53)
    chunks = [6000, 9000, 300, 20_000, 4096, 17_000, 123, 8000]
    n = sum(chunks)
    k1 = rng.integers(0, 3000, size=n).astype(np.int64)
    k2 = [("x", "yy", "zzz", "")[i] for i in rng.integers(0, 4, size=n)]
    v = rng.normal(size=n)
    q = rng.integers(-100, 100, size=n).astype(np.int64)
    flag = rng.integers(0, 3, size=n).astype(np.int64)
    valid_q = np.ones(n, dtype=bool)
    valid_q[40_000:] = rng.random(n - 40_000) > 0.1  # NULL cells only in the LAST shard of a 2-rank cut
    words = None

    def shard(r, w):
        nonlocal words
        t = rt.HipTable(1, chunks, r, w)
        lo = sum(chunks[:t.first_chunk])
        hi = lo + t.local_rows
        t.append_column(1, abi.DT_INT64, k1[lo:hi])
        if w > 1:
            if words is None:
                words = rt.comm_union_strings(sorted(set(k2[lo:hi])))  # ranks agree on the dictionary through the library
            t.append_utf8_column(2, k2[lo:hi], words)
        else:
            t.append_utf8_column(2, k2[lo:hi], sorted(set(k2)))
        t.append_column(3, abi.DT_INT64, q[lo:hi], valid=valid_q[lo:hi] if not valid_q[lo:hi].all() else None)
        t.append_column(4, abi.DT_FLOAT64, v[lo:hi])
        t.append_column(5, abi.DT_INT64, flag[lo:hi])
        if w > 1:
            t.share_metadata()  # statistics + which columns have NULL cells: identical plans on every rank
        return t

    return shard(rank, world)


def _queries(abi):
    A, col = abi.AggregateSpec, abi.col
    D = lambda s: dataclasses.replace(s, distinct=True)
    return {
        # dense GROUP BY on a 3-valued integer key; the SUM argument has NULL cells on one shard only
        "dense": dict(predicate=[abi.Filter(1, abi.Operator.GreaterThan(10))], aggs=[A.count_star(), A.sum(3), A.sum(col(4) * 2.0), A.min(3), A.count(3)], keys=[5], order=True),
        # sort-based GROUP BY: ~12 000 groups straddling the shards
        "sorted": dict(predicate=[abi.Filter(1, abi.Operator.GreaterThan(10))], aggs=[A.count_star(), A.sum(1), A.sum(col(4) * 2.0), A.max(4)], keys=[1, 2], order=False),
        # the same GROUP BY through the shared-image kernel (statistics bound the keys): exact, order-free lanes
        "image": dict(predicate=[abi.Filter(1, abi.Operator.GreaterThan(10))], aggs=[A.count_star(), A.sum(1), A.sum(col(4) * 2.0)], keys=[1, 2], order=True),
        "distinct": dict(predicate=[abi.Filter(1, abi.Operator.GreaterThan(10))], aggs=[D(A.count(1)), D(A.sum(1)), D(A.sum(4)), A.count_star()], keys=[], order=False),
    }


def _join_inputs(rt, abi, tpch, rank, world):
    D = tpch.DATE_1995_03_15
    rows, scale = 60175, 0.01
    li = tpch.gen_lineitem(rows, scale)
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    ot_ = rt.HipTable(2, tpch.chunk_rows(n_ord, 65536))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust, 65536))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    chunks = tpch.chunk_rows(rows, 1000)
    t = rt.HipTable(1, chunks, rank, world)
    lo = sum(chunks[:t.first_chunk])
    for c in ("l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"):
        t.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c][lo:lo + t.local_rows])
    if world > 1:
        t.share_metadata()
    F, O, col = abi.Filter, abi.Operator, abi.col
    return dict(fact=t, fact_filters=[F(tpch.L_SHIPDATE, O.GreaterThan(D))], fact_key=tpch.L_ORDERKEY, dim=ot_,
                dim_filters=[F(tpch.O_ORDERDATE, O.LessThan(D))], dim_key=tpch.O_ORDERKEY, sum_expr=col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT)),
                payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], dim_fk=tpch.O_CUSTKEY, dim2=ct,
                dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)


def _join_group(rt, abi, tpch, rank, world, sharded):
    ji = _join_inputs(rt, abi, tpch, rank, world)
    A, col = abi.AggregateSpec, abi.col
    aggs = [A.sum(ji["sum_expr"]), A.count_star(), A.min(tpch.L_EXTENDEDPRICE), A.avg(tpch.L_EXTENDEDPRICE)]
    jq = rt.JoinGroupBy(ji["fact"], ji["fact_filters"], ji["fact_key"], ji["dim"], ji["dim_filters"], ji["dim_key"], aggs, dim_fk=ji["dim_fk"], dim2=ji["dim2"],
                        dim2_filters=ji["dim2_filters"], dim2_key=ji["dim2_key"])
    jq.launch(0)
    if sharded:
        rt.check(rt.lib().llkv_hip_query_finish_sharded(jq._h, None))
    else:
        jq.finish_only()
    rows, total = jq.result(ji["payload_fields"], [(abi.JOIN_ORDER_AGGREGATE, 1, True), (abi.JOIN_ORDER_AGGREGATE, 2, False), (abi.JOIN_ORDER_KEY, 0, False)], 25)
    return [(r.key, tuple(r.payload), r.group_index, r.values[1].value, r.values[2].value, r.values[0].value, r.values[3].value) for r in rows], total


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    abi, rt, tpch = mod("abi"), mod("runtime"), mod("tpch")
    rt.init(0)  # both ranks on the one GPU of the box
    rt.comm_init_torch(dist, rank, world)
    res = {}
    t = _tables(rt, abi, rank, world)
    for name, q in _queries(abi).items():
        if name == "sorted":
            os.environ["LLKV_HIP_GROUP_NO_IMAGE"] = "1"  # the statistics would admit the shared-image kernel: this case is the sort route's exchange
        pq = rt.PreparedQuery(t, q["predicate"], q["aggs"], q["keys"], q["order"])
        os.environ.pop("LLKV_HIP_GROUP_NO_IMAGE", None)
        pq.launch(0)
        res[name] = _flat(pq.finish_sharded())
        if name == "dense":  # the pipelined form: launch ×2, then all_reduce / submit / collect each
            pq.set_depth(2)
            pq.launch(0); pq.launch(0)
            for _ in range(2):
                pq.all_reduce(0); pq.submit(0)
                assert _flat(pq.collect()) == res[name]
        pq.close()
    ja = rt.JoinAgg(**_join_inputs(rt, abi, tpch, rank, world))
    rows, total = ja.finish_sharded(10)
    res["join"] = ([(r[0], np.float64(r[1]).tobytes(), r[2], r[3], r[4]) for r in rows], total)
    # the range form: every rank selects the dimension rows of its own fact key range and the ranks exchange their boundary
    # runs (one small all-gather inside finish_sharded) instead of the per-group counts
    jr = rt.JoinAgg(ranged=True, **_join_inputs(rt, abi, tpch, rank, world))
    rows, total = jr.finish_sharded(10)
    res["join_ranged"] = ([(r[0], np.float64(r[1]).tobytes(), r[2], r[3], r[4]) for r in rows], total)
    res["join_ranged_bytes"] = jr.exchange_bytes()
    # join → GROUP BY with an aggregate list (llkv_hip_join_groupby_prepare): the prepared GROUP BY of the fact key runs over this
    # rank's shard, finish_sharded all-gathers the partial groups and merges them lane by lane in rank order
    res["join_group"] = _join_group(rt, abi, tpch, rank, world, sharded=True)
    with open(f"{out_path}.{rank}", "wb") as f:
        pickle.dump(res, f)
    rt.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_run_the_sharded_drivers_over_a_host_transport(rt, abi, tpch, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = [pickle.load(open(f"{out}.{r}", "rb")) for r in range(2)]
    assert got[0] == got[1]  # every rank ends with the table-wide answer
    whole = _tables(rt, abi, 0, 1)
    for name, q in _queries(abi).items():
        if name == "sorted":
            os.environ["LLKV_HIP_GROUP_NO_IMAGE"] = "1"
        pw = rt.PreparedQuery(whole, q["predicate"], q["aggs"], q["keys"], q["order"])
        os.environ.pop("LLKV_HIP_GROUP_NO_IMAGE", None)
        if name == "image":
            assert ",2>" in pw.kernel_signature or ",2," in pw.kernel_signature, pw.route_note
        want = _flat(pw.run())
        g = got[0][name]
        assert [k for k, _ in g] == [k for k, _ in want], name  # same groups, same order
        for (_, gv), (_, wv) in zip(g, want):
            for a, (x, y) in enumerate(zip(gv, wv)):
                if isinstance(y, bytes):
                    fx, fy = np.frombuffer(x, np.float64)[0], np.frombuffer(y, np.float64)[0]
                    if name == "sorted":  # one more level of association across the ranks
                        assert abs(fx - fy) <= REL * max(1.0, abs(fy)), (name, a)
                    else:  # dense route: canonical octant fold; DISTINCT: first-appearance order — bit for bit
                        assert x == y, (name, a, fx, fy)
                else:
                    assert x == y, (name, a)
    jw, jtotal = rt.join_groupby_topk(limit=10, **_join_inputs(rt, abi, tpch, 0, 1))
    assert got[0]["join"] == ([(r[0], np.float64(r[1]).tobytes(), r[2], r[3], r[4]) for r in jw], jtotal)
    assert got[0]["join_ranged"] == got[0]["join"] and 0 < got[0]["join_ranged_bytes"] < 8192
    (jg, jg_total), (wg, wg_total) = got[0]["join_group"], _join_group(rt, abi, tpch, 0, 1, sharded=False)
    assert jg_total == wg_total == jtotal and [r[:5] for r in jg] == [r[:5] for r in wg]  # keys, payload, positions, counts, minima: exact
    for g, w in zip(jg, wg):
        assert abs(g[5] - w[5]) <= REL * abs(w[5]) and abs(g[6] - w[6]) <= REL * abs(w[6])  # f64 sums: one more level of association


@pytest.mark.gpu
def test_c_program_runs_sharded_q1_through_the_library_collectives(tmp_path):
    """examples/q1_multi_gpu.c: one process per GPU in plain C99 — communicator id over a file, ncclCommInitRank inside
    the library, table-wide dictionary and metadata, `finish_sharded`.  One rank here (this box has one GPU; the code
    path is the N-rank one: a one-rank RCCL communicator still runs ncclAllReduce on the exchange image)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rust-llkv_amd")
    exe = tmp_path / "q1mg"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "q1_multi_gpu.c"), "-L", libdir, "-lllkv_hip", "-lllkv_tpch", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe), "0", "1", str(tmp_path / "id"), "1500000", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.returncode, out.stdout, out.stderr)
    assert sum(" count " in line for line in out.stdout.splitlines()) == 4  # four groups (RCCL prints its banner to stdout too)
