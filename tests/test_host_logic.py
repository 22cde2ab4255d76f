"""CPU: host logic of the product — C-ABI surface, plan lowering (typing rules of the reference),
shard layout.  No compute calls: there is no GPU here and no CPU fallback in the product."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, mod


@pytest.fixture(scope="module")
def lib():
    return mod("runtime").lib()


def declared_functions(header):
    with open(os.path.join(ROOT, "include", header)) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(llkv_(?:hip|plan|tpch)_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions("llkv_hip.h")
    assert len(names) > 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"libllkv_hip.so does not export: {missing}"
    assert lib.llkv_hip_abi_version() == 1


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: the header compiles as pedantic C99 and a C program links against the library
    and calls through it (what a cgo / Rust FFI binding does).  Without a device the data path says so."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('''#include "llkv_hip.h"
#include <stdio.h>
int main(void) {
  uint64_t rows[1] = {4};
  llkv_hip_table *t = 0;
  if (llkv_hip_abi_version() != LLKV_HIP_ABI_VERSION) return 1;
  if (llkv_hip_table_create(1, rows, 1, 0, 1, &t) != LLKV_OK) return 2;   /* host-only: the chunk layout */
  if (llkv_hip_table_total_rows(t) != 4) return 3;
  if (llkv_hip_device_count() == 0) {
    const long long v[4] = {1, 2, 3, 4};
    const void *chunks[1] = {v};
    if (llkv_hip_table_append_column(t, 1, LLKV_DT_INT64, chunks, 1) != LLKV_NO_DEVICE) return 4;
    if (!llkv_hip_last_error()[0]) return 5;
  }
  llkv_hip_table_free(t);
  puts("ok");
  return 0;
}
''')
    libdir = os.path.join(ROOT, "rust-llkv_amd")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                           "-L", libdir, "-lllkv_hip", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)


def test_batches_export_through_the_arrow_c_data_interface(lib, abi):
    """llkv_hip_batch_export_arrow: a scan batch view becomes an Arrow RecordBatch the consumer owns — imported here
    by pyarrow through the C Data Interface, every storage type, NULL cells, row ids."""
    pa = pytest.importorskip("pyarrow")

    class ArrowSchema(C.Structure):
        pass

    class ArrowArray(C.Structure):
        pass
    ArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                            ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]
    ArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                           ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]
    n = 11
    valid = np.array([1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1], dtype=bool)
    bitmap = np.packbits(valid, bitorder="little")
    i64 = np.arange(n, dtype=np.int64) - 5
    f64 = np.linspace(-1.0, 1.0, n)
    i32 = (np.arange(n, dtype=np.int32) * 7)
    u64 = np.arange(n, dtype=np.uint64) + 2**63
    boo = (np.arange(n) % 3 == 0).astype(np.uint8)
    dec = abi.i128_buffer([(-1) ** k * (10**20 + k) for k in range(n)])
    codes = (np.arange(n) % 3).astype(np.uint8)
    words = [b"alpha", b"", b"gamma-gamma"]
    dictionary = (C.c_char_p * 3)(*words)
    rid = np.arange(n, dtype=np.uint64) * 10
    cols = (abi.CColumnView * 8)()
    specs = [(abi.DT_INT64, i64, bitmap), (abi.DT_FLOAT64, f64, None), (abi.DT_INT32, i32, None), (abi.DT_DATE32, i32, bitmap), (abi.DT_UINT64, u64, None),
             (abi.DT_BOOLEAN, boo, None), (abi.DT_DECIMAL128, dec, bitmap), (abi.DT_UTF8, codes, bitmap)]
    for k, (dt, arr, bm) in enumerate(specs):
        cols[k].dtype = dt
        cols[k].values = arr.ctypes.data
        cols[k].validity = bm.ctypes.data_as(C.POINTER(C.c_uint8)) if bm is not None else None
        cols[k].dictionary = dictionary if dt == abi.DT_UTF8 else None
        cols[k].precision, cols[k].scale = (38, 4) if dt == abi.DT_DECIMAL128 else (0, 0)
    batch = abi.CBatchView(n, 8, cols, rid.ctypes.data_as(C.POINTER(C.c_uint64)))
    names = (C.c_char_p * 8)(b"a", b"b", b"c", b"d", b"e", b"f", b"g", b"h")
    arr, sch = ArrowArray(), ArrowSchema()
    rc = lib.llkv_hip_batch_export_arrow(C.byref(batch), names, C.byref(arr), C.byref(sch))
    assert rc == 0
    rb = pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))  # takes ownership (calls release)
    del i64, f64, u64, dec, codes  # the export owns copies
    assert rb.num_rows == n and rb.schema.names == ["a", "b", "c", "d", "e", "f", "g", "h", "rowid"]
    assert [str(t) for t in rb.schema.types] == ["int64", "double", "int32", "date32[day]", "uint64", "bool", "decimal128(38, 4)", "string", "uint64"]
    mask = lambda vals: [v if ok else None for v, ok in zip(vals, valid)]
    assert rb.column(0).to_pylist() == mask([k - 5 for k in range(n)])
    assert rb.column(1).to_pylist() == np.linspace(-1.0, 1.0, n).tolist()
    assert rb.column(2).to_pylist() == [7 * k for k in range(n)]
    assert [None if v is None else (v - __import__("datetime").date(1970, 1, 1)).days for v in rb.column(3).to_pylist()] == mask([7 * k for k in range(n)])
    assert rb.column(4).to_pylist() == [2**63 + k for k in range(n)]
    assert rb.column(5).to_pylist() == [k % 3 == 0 for k in range(n)]
    from decimal import Decimal
    assert rb.column(6).to_pylist() == mask([Decimal((-1) ** k * (10**20 + k)).scaleb(-4) for k in range(n)])
    assert rb.column(7).to_pylist() == mask([words[k % 3].decode() for k in range(n)])
    assert rb.column(8).to_pylist() == [10 * k for k in range(n)] and rb.column(8).null_count == 0
    assert rb.column(0).null_count == 2
    assert lib.llkv_hip_batch_export_arrow(None, None, C.byref(arr), C.byref(sch)) != 0


def test_generator_library_exports_every_declared_symbol(tpch):
    g = tpch.gen_lib()
    missing = [n for n in declared_functions("llkv_tpch_gen.h") if not hasattr(g, n)]
    assert not missing


def test_data_path_fails_loudly_without_a_device(lib, abi):
    """No silent CPU path: without a bound device prepare/launch return NO_DEVICE."""
    rt = mod("runtime")
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(abi.LlkvError) as e:
        rt.init(0)
    assert e.value.kind == "NoDevice"
    t = rt.HipTable(1, [4])
    with pytest.raises(abi.LlkvError) as e:
        t.append_column(1, abi.DT_INT64, np.arange(4, dtype=np.int64))
    assert e.value.kind == "NoDevice"


def test_generator_is_row_addressable(tpch):
    """Any shard can be generated independently: rows [a,b) equal the slice of rows [0,b)."""
    full = tpch.gen_lineitem(5000, 0.01)
    part = tpch.gen_lineitem(1234, 0.01, row_begin=3000)
    for k in full:
        assert np.array_equal(full[k][3000:4234], part[k]), k
    assert set(np.unique(full["l_returnflag"])) <= {ord("A"), ord("N"), ord("R")}
    assert full["l_quantity"].min() >= 1 and full["l_quantity"].max() <= 50
    assert np.all(np.diff(full["l_orderkey"]) >= 0)
    # discount / tax equal their literal casts (k / 100.0)
    assert set(np.unique(full["l_discount"])) <= {k / 100.0 for k in range(11)}


def test_lowering_of_benchmark_plans_hits_the_aot_catalog(lib, abi, tpch):
    rt = mod("runtime")
    keep = []
    descs = tpch.lineitem_column_descs(tpch.LINEITEM_ROWS["sf10"], keep)
    inc = open(os.path.join(ROOT, "rust-llkv_amd", "csrc", "catalog_entries.inc")).read()
    for name, mk in tpch.QUERIES.items():
        q = mk()
        ts, lanes, bpr = rt.lower_plan(descs, q.predicate, q.aggs, q.keys, q.grouped, order_by_keys=q.order_by_keys)
        assert bpr == q.bytes_per_row  # 16 / 28 / 38 B per row
        assert f'"{ts}"' in inc, f"{name} is not pre-compiled"
    ts, lanes, _ = rt.lower_plan(descs, tpch.q1().predicate, tpch.q1().aggs, tpch.q1().keys, True, order_by_keys=True)
    assert "Keys<6,0,KeyCode<1>,KeyCode<2>>" in ts and lanes == 6 * 6 + 1 and ts.endswith(",4,1>")  # LDS accumulators
    ts, lanes, _ = rt.lower_plan(descs, tpch.q1().predicate, tpch.q1().aggs, tpch.q1().keys, True)
    assert "Keys<6,1,KeyCode<1>,KeyCode<2>>" in ts and lanes == 6 * 7 + 1  # first-appearance order needs the first-row lane


def _desc(abi, cols):
    arr = (abi.CColumnDesc * len(cols))()
    for i, col in enumerate(cols):
        arr[i].field_id, arr[i].dtype, arr[i].rows = col[0], col[1], 1000
        arr[i].nullable = int(len(col) > 2 and col[2])
    return arr


def test_group_by_lowering_forms_and_their_group_limits(lib, abi):
    """The three GROUP BY lowerings over statistics-bounded integer keys: per-thread accumulators (key ranges ≤ 256,
    ≤ 64 groups), the shared-image form (≤ 65 536 dense group ids, ≤ 4 LDS-sized slices, order-free lanes: an f64 sum
    needs bounds from the statistics), and its partitioned form (≤ 2^24 dense ids, no LDS bound, first row always kept)."""
    rt = mod("runtime")
    A = abi.AggregateSpec

    def descs(span):
        d = (abi.CColumnDesc * 3)()
        d[0].field_id, d[0].dtype, d[0].rows, d[0].has_stats, d[0].min_i, d[0].max_i = 1, abi.DT_INT64, 10**7, 1, 1, span
        d[1].field_id, d[1].dtype, d[1].rows, d[1].has_stats, d[1].min_i, d[1].max_i = 2, abi.DT_INT64, 10**7, 1, 1, 50
        d[2].field_id, d[2].dtype, d[2].rows = 3, abi.DT_FLOAT64, 10**7
        d[2].has_fstats, d[2].f_absmax, d[2].f_absmin_nz, d[2].f_all_finite = 1, 105000.0, 900.0, 1
        return d

    aggs = [A.count_star(), A.sum(2), A.sum(3)]
    for span, dense, image, part in ((12, True, True, True), (2526, False, True, True), (2_000_000, False, False, True), ((1 << 24) + 1, False, False, False)):
        for form, ok, tail in ((0, dense, ",1>"), (4, image, ",2"), (12, part, ",3>")):
            if ok:
                ts, lanes, _ = rt.lower_plan(descs(span), None, aggs, [1], True, form=form)
                assert f"Keys<{span},1,KeyInt<0,I64," in ts and tail in ts[-6:], (span, form, ts)
                assert ("SumF64<" in ts) == (form == 0) and ("SumF64Q<" in ts or "SumF64X<" in ts) == (form != 0), ts  # order-free f64 sums off the dense form
            else:
                with pytest.raises(abi.LlkvError) as e:
                    rt.lower_plan(descs(span), None, aggs, [1], True, form=form)
                assert e.value.kind == "Unsupported", (span, form)
    # a whole table's rows may meet in one partition's image: the fixed-point grid of the shared-image form (sized for the
    # rows ONE workgroup sees) does not fit 10^7 rows of this column, the multi-level grids do
    ts_i, _, _ = rt.lower_plan(descs(2526), None, aggs, [1], True, form=4)
    ts_p, _, _ = rt.lower_plan(descs(2526), None, aggs, [1], True, form=12)
    assert "SumF64Q<" in ts_i and "SumF64X<" in ts_p, (ts_i, ts_p)
    # without bounds on the f64 argument there is no order-free sum: only the dense form (and the sort-based route) take it
    d = descs(2526)
    d[2].has_fstats = 0
    with pytest.raises(abi.LlkvError):
        rt.lower_plan(d, None, aggs, [1], True, form=12)


def test_literal_cast_rules(lib, abi):
    """llkv-types/src/literal.rs:364-520: integer columns accept only integer (or scale-0 decimal)
    literals; Date32 filters as i32 and rejects a Date32 literal; f64 accepts ints and decimals."""
    rt = mod("runtime")
    d = _desc(abi, [(1, abi.DT_INT64), (2, abi.DT_FLOAT64), (3, abi.DT_DATE32), (4, abi.DT_INT32)])
    F, O, A, L = abi.Filter, abi.Operator, abi.AggregateSpec, abi.Literal
    cnt = [A.count_star()]
    for bad in (F(1, O.LessThan(24.5)), F(3, O.GreaterThanOrEquals(L.date32(8766))), F(1, O.Equals("x")),
                F(4, O.LessThan(2**40)), F(1, O.LessThan(L.decimal(245, 1)))):
        with pytest.raises(abi.LlkvError) as e:
            rt.lower_plan(d, [bad], cnt)
        assert e.value.kind == "PredicateBuild", bad
    ts, _, _ = rt.lower_plan(d, [F(1, O.LessThan(L.decimal(24, 0))), F(2, O.GreaterThanOrEquals(L.decimal(5, 2))), F(2, O.LessThan(3))], cnt)
    assert ts.startswith("Plan<Cols<I64,F64>,And<Range<Col<0,I64>,0,Nil,2,LitI<0>>,Range<Col<1,F64>,1,LitF<0>,0,Nil>,Range<Col<1,F64>,0,Nil,2,LitF<1>>>")
    with pytest.raises(abi.LlkvError) as e:
        rt.lower_plan(d, [F(9, O.LessThan(1))], cnt)
    assert e.value.kind == "NotFound"


def test_expression_typing_rules(lib, abi):
    """Fast path: everything is cast to the final type first (fast_numeric.rs:69-121); GROUP BY arguments use
    the PlanValue rules where Int∘Int goes through f64 (llkv-executor/src/lib.rs:7338-7389)."""
    rt = mod("runtime")
    d = (abi.CColumnDesc * 3)()
    d[0].field_id, d[0].dtype, d[0].rows = 1, abi.DT_INT64, 10
    d[1].field_id, d[1].dtype, d[1].rows = 2, abi.DT_FLOAT64, 10
    names = (C.c_char_p * 2)(b"x", b"y")
    d[2].field_id, d[2].dtype, d[2].rows, d[2].dict_size, d[2].dictionary = 3, abi.DT_UTF8, 10, 2, names
    A, col = abi.AggregateSpec, abi.col
    ts, _, _ = rt.lower_plan(d, None, [A.sum((col(1) + col(1)) * col(2))])
    assert "SumF64<Bin<3,Bin<1,ToF64<Col<0,I64>>,ToF64<Col<0,I64>>>,Col<1,F64>>>" in ts  # the i64 add happens in f64
    ts, _, _ = rt.lower_plan(d, None, [A.sum(col(1) * 2)])
    assert "SumI64<Bin<3,Col<0,I64>,LitI<0>>>" in ts  # checked i64, exact 96-bit split accumulator
    ts, _, _ = rt.lower_plan(d, None, [A.sum(col(1) * 2), A.sum(col(1) * (1 - col(2)))], keys=[3], grouped=True)
    assert "BinViaF64<3,Col<1,I64>,LitI<0>>" in ts and "Bin<3,ToF64<Col<1,I64>>,Bin<2,ToF64<LitI<1>>,Col<2,F64>>>" in ts
    # Divide leaves the fast path: every node is typed on its own operands, a zero divisor makes the row NULL
    ts, _, _ = rt.lower_plan(d, None, [A.sum((col(1) + col(1)) / col(2)), A.sum(col(1) % 7)])
    assert "IfValid<VE<Div<ToF64<Bin<1,Col<0,I64>,Col<0,I64>>>,Col<1,F64>>>,SumF64<Div<ToF64<Bin<1,Col<0,I64>,Col<0,I64>>>,Col<1,F64>>>>" in ts
    assert "SumI64<Bin<4,Col<0,I64>,LitI<0>>>" in ts  # % stays on the fast path: checked, `% 0` is an error
    ts, _, _ = rt.lower_plan(d, None, [A.sum(col(2) / col(1)), A.sum(col(1) % 3)], keys=[3], grouped=True)
    assert "IfValid<VE<DivPV<0,Col<1,F64>,Col<2,I64>>>,SumF64<DivPV<0,Col<1,F64>,Col<2,I64>>>>" in ts
    assert "IfValid<VE<BinViaF64<4,Col<2,I64>,LitI<0>>>,SumI64<BinViaF64<4,Col<2,I64>,LitI<0>>>>" in ts  # PlanValue x % 0 → NULL
    with pytest.raises(abi.LlkvError) as e:
        rt.lower_plan(d, None, [A.sum(col(1) / col(1))], keys=[3], grouped=True)
    assert e.value.kind == "Unsupported"  # Int / Int in a GROUP BY argument turns Float for i64::MIN / -1
    d32 = _desc(abi, [(1, abi.DT_INT32)])
    with pytest.raises(abi.LlkvError) as e:
        rt.lower_plan(d32, None, [A.sum(1)])
    assert e.value.kind == "InvalidArgumentError" and "not supported for column type Int32" in e.value.message


def test_constant_folding_and_late_columns_in_the_lowering(lib, abi):
    """ScalarEvaluator::simplify before typing (llkv-compute/src/eval.rs:761-791): literal ⊕ literal is one literal, a division
    of constants is gone before the route is picked, a fold that errors is a plan nobody takes.  And the slot order the
    late-materialising kernels rely on: predicate columns first, argument-only columns behind them (Plan<…,0,1,EARLY>)."""
    rt = mod("runtime")
    d = _desc(abi, [(1, abi.DT_INT64), (2, abi.DT_FLOAT64), (3, abi.DT_INT64)])
    A, F, O, col, L = abi.AggregateSpec, abi.Filter, abi.Operator, abi.col, abi.ScalarExpr.literal
    ts, _, _ = rt.lower_plan(d, None, [A.sum(col(1) * (L(2) + 3))])
    assert "SumI64<Bin<3,Col<0,I64>,LitI<0>>>" in ts and "LitI<1>" not in ts
    ts, _, _ = rt.lower_plan(d, None, [A.sum(col(1) * (L(1) / 4))])
    assert "Div<" not in ts and "SumI64<Bin<3,Col<0,I64>,LitI<0>>>" in ts  # times the folded 0, on the fast path
    ts, _, _ = rt.lower_plan(d, None, [A.sum(col(2) * (L(1) / 4.0) + (L(3) - 1))])
    assert "LitF<1>" in ts and "LitF<2>" not in ts and "Div<" not in ts and "IfValid<" not in ts  # two literals are left of the four; nothing can be NULL
    for bad in (col(1) + (L(2**62) + 2**62), col(1) + (L(5) % 0), col(1) * (L(-2**63) / -1), col(1) + (L(1) / 0)):
        with pytest.raises(abi.LlkvError) as e:
            rt.lower_plan(d, None, [A.sum(bad)])
        assert e.value.kind == "Unsupported", bad
    # late columns: field 3 feeds the predicate (slot 0), fields 1 and 2 only the arguments
    ts, _, _ = rt.lower_plan(d, [F(3, O.LessThan(7))], [A.sum(col(1) * 2), A.sum(2)])
    assert ts.startswith("Plan<Cols<I64,I64,F64>,And<Range<Col<0,I64>") and ts.endswith(",0,1,1>"), ts
    ts, _, _ = rt.lower_plan(d, [F(3, O.LessThan(7))], [A.sum(3), A.count_star()])
    assert ts.endswith(",0>"), ts  # nothing is argument-only: the eager form
    ts, _, _ = rt.lower_plan(d, None, [A.sum(col(1) * 2)])
    assert ts.endswith(",0>"), ts  # no predicate: every row passes


def test_compare_lowering_follows_the_common_type_rules(lib, abi):
    """Expr::Compare (llkv-scan/src/predicate.rs:333-396): column ⋈ literal is the leaf filter; anything else
    is evaluated per side and coerced with get_common_type (llkv-compute/src/kernels.rs:179-242)."""
    rt = mod("runtime")
    d = _desc(abi, [(1, abi.DT_INT64), (2, abi.DT_FLOAT64), (3, abi.DT_INT32), (4, abi.DT_UINT64), (5, abi.DT_UINT32)])
    E, col, cnt = abi.Expr, abi.col, [abi.AggregateSpec.count_star()]
    F, O = abi.Filter, abi.Operator
    # simple_compare_filter: same plan as the leaf predicate, literal ⋈ column flips the operator
    leaf, _, _ = rt.lower_plan(d, E.pred(F(3, O.GreaterThan(7))), cnt)
    assert rt.lower_plan(d, E.compare(col(3), abi.CMP_GT, 7), cnt)[0] == leaf
    assert rt.lower_plan(d, E.compare(7, abi.CMP_LT, col(3)), cnt)[0] == leaf
    # <> is never a leaf: Int32 column vs Int64 literal → Int64 compare
    ts, _, _ = rt.lower_plan(d, E.compare(col(3), abi.CMP_NOT_EQ, 7), cnt)
    assert "Cmp<2,ToI64<Col<0,I32>>,LitI<0>>" in ts
    # the reference's own case: UInt64 + Int32 is Float64, so the Int64 literal is compared as Float64
    ts, _, _ = rt.lower_plan(d, E.compare(col(4) + col(3), abi.CMP_GT, 220), cnt)
    assert "Cmp<5,Bin<1,ToF64<Col<0,U64>>,ToF64<Col<1,I32>>,1>,ToF64<LitI<0>>>" in ts  # ",1": a computed NaN keeps the host's sign (totalOrder)
    # signed ⋈ unsigned: 64 bits wide → Float64, 32 bits → Int64; unsigned ⋈ unsigned stays unsigned
    assert "Cmp<3,ToF64<Col<0,I64>>,ToF64<Col<1,U64>>>" in rt.lower_plan(d, E.compare(col(1), abi.CMP_LT, col(4)), cnt)[0]
    assert "Cmp<3,ToF64<ToI64<Col<0,U32>>>,ToF64<Col<1,I64>>>" in rt.lower_plan(d, E.compare(col(5), abi.CMP_LT, col(1)), cnt)[0]
    assert "Cmp<3,ToI64<Col<0,I32>>,ToI64<Col<1,U32>>>" in rt.lower_plan(d, E.compare(col(3), abi.CMP_LT, col(5)), cnt)[0]
    assert "Cmp<6,Col<0,U64>,ToI64<Col<1,U32>>>" in rt.lower_plan(d, E.compare(col(4), abi.CMP_GT_EQ, col(5)), cnt)[0]
    assert "Cmp<1,Bin<3,Col<0,I64>,LitI<0>>,ToI64<Col<1,I32>>>" in rt.lower_plan(d, E.compare(col(1) * 3, abi.CMP_EQ, col(3)), cnt)[0]
    # Int32-only arithmetic stays Int32: every node checked against 32 bits, the side compares as an Int32 does
    assert "Cmp<3,Fit32<Bin<1,ToI64<Col<0,I32>>,ToI64<Col<0,I32>>>,1>,Col<1,I64>>" in rt.lower_plan(d, E.compare(col(3) + col(3), abi.CMP_LT, col(1)), cnt)[0]
    assert "Cmp<3,ToF64<Fit32<Bin<3,ToI64<Col<0,U32>>,ToI64<Col<0,U32>>>,0>>,ToF64<Col<1,I64>>>" in rt.lower_plan(d, E.compare(col(5) * col(5), abi.CMP_LT, col(1)), cnt)[0]
    # no field at all: evaluate_constant_compare — TRUE keeps every row, FALSE / NULL none (r04)
    L, NULL = abi.ScalarExpr.literal, abi.ScalarExpr.literal(abi.Literal.of(None))
    assert ",True," in rt.lower_plan(d, E.compare(L(1) + 2, abi.CMP_LT, 4), cnt)[0]
    assert ",False," in rt.lower_plan(d, E.compare(L(1) + 2, abi.CMP_GT, 4.5), cnt)[0]
    assert ",False," in rt.lower_plan(d, E.not_(E.compare(NULL, abi.CMP_EQ, 1)), cnt)[0]
    # a side that IS the NULL literal: nothing matches, nothing is determined — but the other side's arithmetic is still evaluated
    assert ",False," in rt.lower_plan(d, E.compare(col(1), abi.CMP_NOT_EQ, NULL), cnt)[0]
    assert "ErrOnly<Bin<3,Col<0,I64>,LitI<0>>>" in rt.lower_plan(d, E.compare(col(1) * 3, abi.CMP_GT, NULL), cnt)[0]
    for bad in (E.compare(col(1) + NULL, abi.CMP_LT, 4),):  # a NULL literal inside a side's arithmetic
        with pytest.raises(abi.LlkvError) as e:
            rt.lower_plan(d, bad, cnt)
        assert e.value.kind == "Unsupported"


def test_null_cells_lower_to_validity_masks_and_domains(lib, abi):
    """A NULL cell never matches a leaf (table.rs:1241-1244); NOT is taken inside the child's domain
    (predicate.rs:167-186, program.rs:447-520); accumulators skip NULL arguments and count what they saw."""
    rt = mod("runtime")
    d = _desc(abi, [(1, abi.DT_INT64, True), (2, abi.DT_FLOAT64, True), (3, abi.DT_INT64)])
    E, F, O, A, col = abi.Expr, abi.Filter, abi.Operator, abi.AggregateSpec, abi.col
    cnt = [A.count_star()]
    ts = rt.lower_plan(d, E.pred(F(1, O.LessThan(5))), cnt)[0]
    assert "Cols<U8,I64>,And<Valid<0>,Range<Col<1,I64>,0,Nil,2,LitI<0>>>" in ts
    # NOT: domain = rows where the field is present; a NULL-free column keeps the plain complement
    assert "And<Valid<0>,Not<And<Valid<0>,Range<" in rt.lower_plan(d, E.not_(F(1, O.LessThan(5))), cnt)[0]
    assert ",Not<Range<Col<0,I64>" in rt.lower_plan(d, E.not_(F(3, O.LessThan(5))), cnt)[0]
    # Or → union of the domains, And → intersection; a NULL-free child makes the union every row
    ts = rt.lower_plan(d, E.not_(E.any_of([F(1, O.LessThan(5)), F(2, O.GreaterThan(1.0))])), cnt)[0]
    assert "And<Or<Valid<0>,Valid<2>>,Not<Or<" in ts
    ts = rt.lower_plan(d, E.not_(E.any_of([F(1, O.LessThan(5)), F(3, O.GreaterThan(1))])), cnt)[0]
    assert ",Not<Or<And<Valid<0>,Range<" in ts and "And<Or<" not in ts
    ts = rt.lower_plan(d, E.not_(E.all_of([F(1, O.LessThan(5)), F(2, O.GreaterThan(1.0))])), cnt)[0]
    assert "And<And<Valid<0>,Valid<2>>,Not<And<" in ts
    # IS NULL / IS NOT NULL are row-universe algebra on the mask; unbounded ranges keep NULL rows
    assert ",Not<Valid<0>>," in rt.lower_plan(d, E.pred(F(1, O.IsNull)), cnt)[0]
    assert ",Valid<0>,Keys" in rt.lower_plan(d, E.pred(F(1, O.IsNotNull)), cnt)[0]
    assert ",True,Keys" in rt.lower_plan(d, E.pred(F(1, O.Range(abi.Bound.Unbounded, abi.Bound.Unbounded))), cnt)[0]
    # compares are determined where every referenced field is present
    assert "Cols<I64,F64,U8,U8>,Cmp<5,ToF64<Col<0,I64>>,Col<1,F64>,And<Valid<2>,Valid<3>>>" in rt.lower_plan(d, E.compare(col(1), abi.CMP_GT, col(2)), cnt)[0]
    # aggregates
    ts, lanes, _ = rt.lower_plan(d, None, [A.count(1), A.count_nulls(1), A.sum(1), A.avg(2), A.sum(col(1) * col(2)), A.sum(3)])
    assert "CountIf<Valid<0>>" in ts and "IfValid<Valid<0>,SumI64<Col<1,I64>>>" in ts and "IfValid<Valid<2>,SumF64<Col<3,F64>>>" in ts
    assert "IfValid<VE<Bin<3,ToF64<ColN<1,I64,0>>,ColN<3,F64,2>>>,SumF64<Bin<3,ToF64<ColN<1,I64,0>>,ColN<3,F64,2>>>>" in ts and ",SumI64<Col<4,I64>>>" in ts
    # a key column with NULL cells: NULL is one more group (GroupKeyValue::Null)
    dk = _desc(abi, [(1, abi.DT_INT64, True), (2, abi.DT_FLOAT64, True)])
    dk[0].has_stats, dk[0].min_i, dk[0].max_i = 1, 10, 13
    ts, lanes, _ = rt.lower_plan(dk, None, [A.count_star(), A.sum(2)], keys=[1], grouped=True)
    assert "Keys<5,1,KeyOrNull<Valid<1>,KeyInt<0,I64,LitI<0>>,4>>" in ts and lanes == 5 * 4 + 1  # rows, first row, sum, non-NULL count


def test_utf8_ordering_predicates_become_code_sets(lib, abi):
    """Utf8 columns are 1-byte dictionary codes in HBM: an ordering predicate is evaluated once per dictionary
    string on the host (str::cmp, llkv-expr/src/typed_predicate.rs:171-185) and shipped as a 256-bit code set."""
    rt = mod("runtime")
    d = (abi.CColumnDesc * 1)()
    names = (C.c_char_p * 4)(b"pear", b"apple", b"fig", b"zebra")  # codes in first-appearance order, not sorted
    d[0].field_id, d[0].dtype, d[0].rows, d[0].dict_size, d[0].dictionary = 1, abi.DT_UTF8, 10, 4, names
    F, O, B, cnt = abi.Filter, abi.Operator, abi.Bound, [abi.AggregateSpec.count_star()]
    ts, _, _ = rt.lower_plan(d, [F(1, O.GreaterThan("fig"))], cnt)
    assert "InMask<Col<0,U8>,LitU<0>,LitU<1>,LitU<1>,LitU<1>>" in ts  # (the three empty words of the set share one literal slot)
    assert "And<False>" in rt.lower_plan(d, [F(1, O.LessThan("apple"))], cnt)[0]      # no dictionary string qualifies
    assert "InMask<" in rt.lower_plan(d, [F(1, O.Range(B.Included("b"), B.Excluded("q")))], cnt)[0]
    with pytest.raises(abi.LlkvError) as e:
        rt.lower_plan(d, [F(1, O.LessThan(3))], cnt)
    assert e.value.kind == "PredicateBuild"


def test_in_list_and_is_null_expression_lowering(lib, abi):
    """Expr::InList / Expr::IsNull over scalar expressions (llkv-scan/src/predicate.rs:249-331,443-560)."""
    rt = mod("runtime")
    d = _desc(abi, [(1, abi.DT_INT64, True), (2, abi.DT_FLOAT64), (3, abi.DT_INT64)])
    E, col, cnt = abi.Expr, abi.col, [abi.AggregateSpec.count_star()]
    ts = rt.lower_plan(d, E.in_list(col(3), [1, col(2), 4]), cnt)[0]  # the Float64 item widens the target for the items after it
    assert "Or<Cmp<1,LitI<0>,Col<0,I64>>,Cmp<1,Col<1,F64>,ToF64<Col<0,I64>>>,Cmp<1,ToF64<LitI<1>>,ToF64<Col<0,I64>>>>" in ts
    ts = rt.lower_plan(d, E.in_list(col(1), [7], negated=True), cnt)[0]
    assert "Cols<I64,U8>,And<Valid<1>,Not<Cmp<1,LitI<0>,Col<0,I64>>>>" in ts
    assert rt.lower_plan(d, E.is_null(col(1)), cnt)[0] == rt.lower_plan(d, E.pred(abi.Filter(1, abi.Operator.IsNull)), cnt)[0]
    ts = rt.lower_plan(d, E.is_null(col(1) + col(3)), cnt)[0]  # a never-NULL field makes every row part of the scan
    assert ",Not<VE<Bin<1,ColN<1,I64,0>,Col<2,I64>>>>,Keys" in ts
    ts = rt.lower_plan(d, E.is_null(col(3) / col(3), negated=True), cnt)[0]
    assert ",VE<Div<Col<0,I64>,Col<0,I64>>>,Keys" in ts
    # without a column (r04): decided on the host — every row of the table or none
    Lit = abi.ScalarExpr.literal
    assert ",False," in rt.lower_plan(d, E.in_list(Lit(3), [1, 2]), cnt)[0] and ",True," in rt.lower_plan(d, E.in_list(Lit(3), [1, Lit(1) + 2]), cnt)[0]
    assert ",False," in rt.lower_plan(d, E.is_null(Lit(1) + 2), cnt)[0] and ",True," in rt.lower_plan(d, E.is_null(Lit(1) + 2, negated=True), cnt)[0]
    for bad in (E.in_list(col(3), [col(3) / 2]),):
        with pytest.raises(abi.LlkvError) as e:
            rt.lower_plan(d, bad, cnt)
        assert e.value.kind == "Unsupported"


def test_int_sum_uses_statistics_to_exclude_overflow(lib, abi):
    rt = mod("runtime")
    d = (abi.CColumnDesc * 1)()
    d[0].field_id, d[0].dtype, d[0].rows, d[0].has_stats, d[0].min_i, d[0].max_i = 1, abi.DT_INT64, 60_000_000, 1, 1, 50
    ts, _, _ = rt.lower_plan(d, None, [abi.AggregateSpec.sum(1)])
    assert "SumI64Fast<Col<0,I64>>" in ts
    d[0].max_i = 2**62
    ts, _, _ = rt.lower_plan(d, None, [abi.AggregateSpec.sum(1)])
    assert "SumI64<Col<0,I64>>" in ts


@pytest.mark.parametrize("n_chunks", [0, 1, 7, 8, 458, 916, 3664])
def test_shard_layout_partitions_chunks(lib, n_chunks):
    """Every chunk belongs to exactly one octant and one rank; shards are contiguous; world 1/2/4/8 shard
    boundaries are octant boundaries (so partial states never straddle ranks)."""
    dist = mod("dist")
    begin1, _ = dist.shard_layout(lib, n_chunks, 1)
    assert begin1[0] == 0 and begin1[8] == n_chunks and all(b <= c for b, c in zip(begin1, begin1[1:]))
    rt = mod("runtime")
    for world in (1, 2, 4, 8):
        begin, owner = dist.shard_layout(lib, n_chunks, world)
        assert begin == begin1
        covered = []
        for rank in range(world):
            t = rt.HipTable(1, [100] * n_chunks, rank, world)
            octs = [o for o in range(8) if owner[o] == rank]
            assert octs == list(range(rank * 8 // world, (rank + 1) * 8 // world))
            assert t.first_chunk == begin[octs[0]] and t.n_local_chunks == begin[octs[-1] + 1] - begin[octs[0]]
            covered += list(range(t.first_chunk, t.first_chunk + t.n_local_chunks))
            assert t.total_rows == 100 * n_chunks and t.local_rows == 100 * t.n_local_chunks
        assert covered == list(range(n_chunks))


def test_arr0_chunk_reader(lib, abi):
    """`ARR0` header (llkv-column-map/src/serialization.rs:41-140): magic, layout, PrimType code, len, extra_a/b."""
    rt = mod("runtime")
    blob = rt.arr0_serialize(abi.DT_INT64, np.arange(5, dtype=np.int64))
    d = rt.arr0_describe(blob)
    assert (d.layout, d.type_code, d.dtype, d.len, d.payload_offset, d.values_len) == (0, 6, abi.DT_INT64, 5, 24, 40)
    d = rt.arr0_describe(rt.arr0_serialize(abi.DT_DATE32, np.arange(3, dtype=np.int32)))
    assert (d.type_code, d.dtype, d.values_len) == (16, abi.DT_DATE32, 12)
    d = rt.arr0_describe(rt.arr0_serialize(abi.DT_UTF8, ["ab", "", "cde"]))
    assert (d.layout, d.type_code, d.dtype, d.len, d.offsets_len, d.values_len, d.values_offset) == (2, 12, abi.DT_UTF8, 3, 16, 5, 40)
    for bad in (b"ARR1" + blob[4:], blob[:20], blob + b"x"):
        with pytest.raises(abi.LlkvError) as e:
            rt.arr0_describe(bad)
        assert e.value.kind == "Internal"
    d = rt.arr0_describe(b"ARR0" + bytes([0, 18, 15, 2]) + (2).to_bytes(8, "little") + (32).to_bytes(4, "little") + bytes(4) + bytes(32))
    assert (d.dtype, d.len, d.values_len) == (abi.DT_DECIMAL128, 2, 32)  # precision 15 / scale 2 ride in header bytes 6 / 7 (serialization.rs:282-296)
    d = rt.arr0_describe(rt.arr0_serialize(abi.DT_BOOLEAN, [True, False, True, True, False, False, True, False, True]))
    assert (d.type_code, d.dtype, d.len, d.values_len) == (15, abi.DT_BOOLEAN, 9, 2)  # an arrow bit buffer
    d = rt.arr0_describe(rt.arr0_serialize(abi.DT_DECIMAL128, [10**20, -5], precision=38, scale=4))
    assert (d.type_code, d.dtype, d.len, d.values_len) == (18, abi.DT_DECIMAL128, 2, 32)
    for code in (7, 8, 9, 10, 17):  # Int16 / Int8 / UInt16 / UInt8 / Date64: no storage type of theirs on this path
        assert rt.arr0_describe(b"ARR0" + bytes([0, code, 0, 0]) + (1).to_bytes(8, "little") + (8).to_bytes(4, "little") + bytes(4) + bytes(8)).dtype == -1


def test_dense_row_runs(lib):
    """store/scan/filter.rs:1510-1582: chunks must each span exactly row_count ids and follow one another."""
    rt = mod("runtime")
    assert rt.dense_row_runs([(131072, 0, 131071), (131072, 131072, 262143), (7, 262144, 262150)]) == (True, 0)
    assert rt.dense_row_runs([(4, 1, 4)]) == (True, 1)
    assert rt.dense_row_runs([]) == (True, 0)
    assert rt.dense_row_runs([(4, 1, 5)])[0] is False          # a hole inside a chunk
    assert rt.dense_row_runs([(4, 0, 3), (4, 5, 8)])[0] is False  # a gap between chunks
    assert rt.dense_row_runs([(4, 0, 3), (0, 0, 0), (2, 4, 5)])[0] is True  # empty chunks are skipped
    assert rt.dense_row_runs([(4, 3, 0)])[0] is False


def test_qualification_value_parsing_and_tolerance():
    """llkv-tpch/src/qualification.rs:947-993 (its in-file tests) restated."""
    from decimal import Decimal
    q = mod("qualify")
    assert q.parse_expected_value("NULL", "integer") == ("null", None)
    assert q.parse_expected_value("hello", "string") == ("string", "hello")
    assert q.parse_expected_value("42", "integer") == ("int", 42)
    assert q.parse_expected_value("12.50", "decimal") == ("decimal", Decimal("12.5"))
    tag, v = q.parse_expected_value("0.125", "float")
    assert tag == "float" and abs(v - 0.125) < 1e-9
    dec, flt = ("decimal", Decimal("1.2345")), ("float", 1.2345)
    assert q.values_equal(dec, flt, "float") and q.values_equal(flt, dec, "float")
    assert q.values_equal(("float", 1.0), ("float", 1.0 + 5e-10), "float")
    assert not q.values_equal(("float", 1.0), ("float", 1.0 + 2e-9), "float")  # absolute tolerance 1e-9
    assert not q.values_equal(("int", 1), ("float", 1.0), "integer")
    d = q.diff_rows([[("int", 1)]], [[("int", 2)]], ["integer"])
    assert len(d.missing) == 1 and len(d.extra) == 1
    d = q.diff_rows([[("int", 1)], [("int", 2)]], [[("int", 2)], [("int", 1)]], ["integer"])  # order-insensitive
    assert d.ok
    with pytest.raises(ValueError):
        q.kind_from_token("xyz")
    rows = q.parse_answer_set("l_returnflag|cnt\nA |  3\n\nN|NULL\n", ["string", "integer"])
    assert rows == [[("string", "A"), ("int", 3)], [("string", "N"), ("null", None)]]


def test_max_threads_follows_the_reference_pool_rule(monkeypatch):
    """configured_thread_count (llkv-threading/src/lib.rs:22-31): LLKV_MAX_THREADS trimmed and parsed as usize; zero,
    a negative or unparsable value falls back to the detected parallelism — the reference's own test
    (`env_override_zero_defaults`, :97-115) asserts the zero case."""
    rt = mod("runtime")
    monkeypatch.delenv("LLKV_MAX_THREADS", raising=False)
    detected = rt.max_threads()
    assert 1 <= detected <= (os.cpu_count() or 1)
    for raw, want in [("0", detected), ("3", 3), (" 5 ", 5), ("+2", 2), ("abc", detected), ("-2", detected), ("", detected), ("2.5", detected), ("1000", 1000)]:
        monkeypatch.setenv("LLKV_MAX_THREADS", raw)
        assert rt.max_threads() == want, raw


def test_route_selection_mirrors_the_executor_dispatch():
    """llkv_hip_select_route = the if-chain of QueryExecutor::execute_select_with_filter
    (llkv-executor/src/lib.rs:531-561), plus whether the GPU path has an entry point for the shape."""
    route = mod("runtime").select_route
    # compound beats everything, then "no table", then GROUP BY (single table → group-by route, several → cross product)
    assert route(has_compound=1, n_tables=2, n_group_by=1)[:2] == ("compound", False)
    assert route(n_tables=0, n_aggregates=1)[:2] == ("no_table", False)
    assert route(n_tables=1, n_group_by=2, n_aggregates=3)[:2] == ("group_by", True)          # Q1
    assert route(n_tables=3, n_group_by=3, has_computed_aggregates=1)[:2] == ("cross_product", True)  # Q3
    assert route(n_tables=2, n_joins=1)[:2] == ("cross_product", True)                        # join_stream
    assert route(n_tables=1, n_aggregates=1)[:2] == ("aggregates", True)                      # configs[0]
    assert route(n_tables=1, has_computed_aggregates=1)[:2] == ("computed_aggregates", True)  # Q6
    assert route(n_tables=1)[:2] == ("projection", True)                                      # scan_stream
    # plain aggregates win over computed ones (:552 before :555)
    assert route(n_tables=1, n_aggregates=1, has_computed_aggregates=1)[0] == "aggregates"
    # SQL breadth the GPU path leaves to the CPU routes: the route is still reported
    name, served, why = route(n_tables=1, n_group_by=1, has_having=1)
    assert (name, served) == ("group_by", False) and "HAVING" in why
    assert route(n_tables=1, has_distinct=1)[:2] == ("projection", False)
    assert route(n_tables=1, n_aggregates=1, has_distinct=1)[:2] == ("aggregates", True)      # DISTINCT aggregates are on the path
    assert route(n_tables=4, n_group_by=1)[:2] == ("cross_product", False)
    assert route(n_tables=1, has_scalar_subqueries=1)[1] is False


def test_query_rendering_with_default_substitution_parameters(abi, tpch):
    """render_tpch_query / build_parameter_values (llkv-tpch/src/queries.rs:60-121,203-232): the defaults of the
    toolkit's varsub.c (the TPC-H validation values), caller overrides by 1-based placeholder index, an unknown
    placeholder is an error; what is rendered here is the plan, not SQL."""
    q = mod("qualify")
    assert q.render_parameters(1) == ["90"] and q.render_parameters(6) == ["1994-01-01", "0.06", "24"] and q.render_parameters(3) == ["BUILDING", "1995-03-15"]
    assert q.render_parameters(6, {2: "0.05"}) == ["1994-01-01", "0.05", "24"]
    with pytest.raises(ValueError):
        q.render_parameters(6, {4: "x"})
    with pytest.raises(ValueError):
        q.render_parameters(2)
    # the defaults reproduce the benchmark plans of tpch.py (BASELINE.json configs[1], [2], [4])
    def lit(x):
        return None if x is None else (x.tag, x.int_value, x.float_value, x.string)

    def bound(b):
        return None if b is None else (b.kind, lit(b.value))

    def same_pred(a, b):
        key = lambda f: (f.field_id, f.op.kind, lit(f.op.value), bound(f.op.lower), bound(f.op.upper))
        return [key(f) for f in a] == [key(f) for f in b]
    assert same_pred(q.render_query(tpch, abi, 1).predicate, tpch.q1().predicate)
    assert same_pred(q.render_query(tpch, abi, 6).predicate, tpch.q6().predicate)
    q3 = q.render_query(tpch, abi, 3)
    assert q3["fact_filters"][0].op.value.int_value == tpch.DATE_1995_03_15 and q3["dim2_filters"][0].op.value.string == "BUILDING" and q3["limit"] == 10
    # an override moves the rendered bounds
    alt = q.render_query(tpch, abi, 6, {1: "1995-01-01", 3: "25"})
    assert alt.predicate[0].op.lower.value.int_value == 9131 and alt.predicate[0].op.upper.value.int_value == 9496 and alt.predicate[2].op.value.int_value == 25


def test_decimal_from_f64_keeps_fifteen_significant_digits():
    """extract_decimal (qualification.rs:532-540) reads a Float64 result cell through rust_decimal's Decimal::from_f64,
    which drops the excess binary precision: 15 significant digits.  A `sum`-kind column is then compared EXACTLY."""
    from decimal import Decimal
    q = mod("qualify")
    assert q.decimal_from_f64(0.1) == Decimal("0.1")
    assert q.decimal_from_f64(37734107.00000001) == Decimal("37734107")
    assert q.decimal_from_f64(56586554400.730011) == Decimal("56586554400.73")
    assert q.decimal_from_f64(-2.5e-7) == Decimal("-2.5E-7")
    assert q.values_equal(("decimal", Decimal("56586554400.73")), q.engine_value(56586554400.73001, "decimal"), "decimal")
    assert not q.values_equal(("decimal", Decimal("56586554400.73")), q.engine_value(56586554400.74, "decimal"), "decimal")
    text = q.format_answer_set(["k", "s", "a", "n"], [["A", 56586554400.730011, 25.5, 7]], ["string", "decimal", "float", "integer"])
    assert text == "k|s|a|n\nA|56586554400.73|25.5|7\n"
    assert q.parse_answer_set(text, ["string", "decimal", "float", "integer"]) == [[("string", "A"), ("decimal", Decimal("56586554400.73")), ("float", 25.5), ("int", 7)]]


def test_string_to_number_coercion_follows_the_rust_float_grammar(lib):
    """SUM / AVG / TOTAL / MIN / MAX over a Utf8 column accumulate `s.trim().parse::<f64>().unwrap_or(0.0)`
    (llkv-aggregate/src/lib.rs:426-434).  The GPU path parses every dictionary entry once on the host; the oracle parses
    every cell.  Both follow Rust's grammar, not strtod's: no hexadecimal forms, "1." and ".5" but not ".", inf /
    infinity / nan in any case, Unicode white space trimmed, anything else counts as 0."""
    import math
    from oracle import oracle as orc
    abi = mod("abi")
    lib.llkv_plan_parse_numeric.restype = C.c_double
    lib.llkv_plan_parse_numeric.argtypes = [C.c_char_p]
    cases = [("12", 12.0), (" 12 ", 12.0), ("3.5e2", 350.0), ("-7.25", -7.25), ("+4", 4.0), ("1.", 1.0), (".5", 0.5), (".", 0.0), ("", 0.0), ("abc", 0.0),
             ("0x10", 0.0), ("1e", 0.0), ("1e+", 0.0), ("1 2", 0.0), ("12abc", 0.0), ("1E3", 1000.0), ("1e-2", 0.01), ("inf", math.inf), ("-Infinity", -math.inf),
             ("+INF", math.inf), ("infinit", 0.0), ("\u00a07\u2003", 7.0), ("\t\n 8 \r\x0b\x0c", 8.0), ("1_000", 0.0), ("٣", 0.0), ("1e400", math.inf),
             ("0.1", 0.1), ("123456789012345678901234567890", 1.2345678901234568e29)]
    for text, want in cases:
        got = lib.llkv_plan_parse_numeric(text.encode())
        assert got == want, (text, got, want)
        t = orc.OracleTable(1).add(1, abi.DT_UTF8, [text])
        v = orc.aggregate(t, None, [abi.AggregateSpec.total(1)])[0].value
        assert v == want, (text, v, want)
    assert math.isnan(lib.llkv_plan_parse_numeric(b"NaN")) and math.isnan(lib.llkv_plan_parse_numeric(b" -nan "))
    t = orc.OracleTable(1).add(1, abi.DT_UTF8, ["nan"])
    assert math.isnan(orc.aggregate(t, None, [abi.AggregateSpec.total(1)])[0].value)
