"""TPC-H harness side: schema, deterministic synthetic data (libllkv_tpch.so) and the
benchmark queries as plans, rendered the way the reference's harness renders them
(llkv-tpch/src/queries.rs:60-121) with literal bounds so the leaf-predicate route is taken
(SURVEY.md §8d).  Column types follow llkv-sql/src/lib.rs:25-28.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import abi
from .abi import (AggregateSpec, Bound, Filter, Operator, col, DT_DATE32, DT_FLOAT64, DT_INT64, DT_UTF8)

_HERE = os.path.dirname(os.path.abspath(__file__))

SEED = 20240607
CHUNK_ROWS = 131072  # llkv-column-map/src/store/constants.rs:22 (1 MiB of 8-byte values)
LINEITEM_ROWS = {"sf0.01": 60175, "sf1": 6001215, "sf10": 59986052}
SCALE = {"sf0.01": 0.01, "sf1": 1.0, "sf10": 10.0}

DATE_1994_01_01, DATE_1995_01_01, DATE_1995_03_15, DATE_1995_06_17, DATE_1998_09_02 = 8766, 9131, 9204, 9298, 10471

# field ids (FieldId = u32, user fields start at 1)
L_ORDERKEY, L_PARTKEY, L_SUPPKEY, L_LINENUMBER, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, \
    L_RETURNFLAG, L_LINESTATUS, L_SHIPDATE, L_COMMITDATE, L_RECEIPTDATE = range(1, 14)
O_ORDERKEY, O_CUSTKEY, O_ORDERDATE, O_SHIPPRIORITY = range(1, 5)
C_CUSTKEY, C_MKTSEGMENT = range(1, 3)

LINEITEM_SCHEMA = {
    "l_orderkey": (L_ORDERKEY, DT_INT64), "l_partkey": (L_PARTKEY, DT_INT64), "l_suppkey": (L_SUPPKEY, DT_INT64),
    "l_linenumber": (L_LINENUMBER, DT_INT64), "l_quantity": (L_QUANTITY, DT_INT64),
    "l_extendedprice": (L_EXTENDEDPRICE, DT_FLOAT64), "l_discount": (L_DISCOUNT, DT_FLOAT64),
    "l_tax": (L_TAX, DT_FLOAT64), "l_returnflag": (L_RETURNFLAG, DT_UTF8), "l_linestatus": (L_LINESTATUS, DT_UTF8),
    "l_shipdate": (L_SHIPDATE, DT_DATE32), "l_commitdate": (L_COMMITDATE, DT_DATE32),
    "l_receiptdate": (L_RECEIPTDATE, DT_DATE32),
}
ORDERS_SCHEMA = {"o_orderkey": (O_ORDERKEY, DT_INT64), "o_custkey": (O_CUSTKEY, DT_INT64),
                 "o_orderdate": (O_ORDERDATE, DT_DATE32), "o_shippriority": (O_SHIPPRIORITY, DT_INT64)}
CUSTOMER_SCHEMA = {"c_custkey": (C_CUSTKEY, DT_INT64), "c_mktsegment": (C_MKTSEGMENT, DT_UTF8)}
SEGMENTS = ["AUTOMOBILE", "BUILDING", "FURNITURE", "HOUSEHOLD", "MACHINERY"]

_NP = {DT_INT64: np.int64, DT_FLOAT64: np.float64, DT_DATE32: np.int32, DT_UTF8: np.uint8}

_lib = None


def gen_lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libllkv_tpch.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(path)
        _lib.llkv_tpch_orders_for_lineitems.restype = C.c_uint64
        _lib.llkv_tpch_orders_for_lineitems.argtypes = [C.c_uint64]
        _lib.llkv_tpch_customers_for_scale.restype = C.c_uint64
        _lib.llkv_tpch_customers_for_scale.argtypes = [C.c_double]
        _lib.llkv_tpch_gen_lineitem.restype = None
        _lib.llkv_tpch_gen_lineitem.argtypes = [C.c_uint64, C.c_double, C.c_uint64, C.c_uint64] + [C.c_void_p] * 13 + [C.c_int32]
        _lib.llkv_tpch_gen_orders.restype = None
        _lib.llkv_tpch_gen_orders.argtypes = [C.c_uint64, C.c_double, C.c_uint64, C.c_uint64] + [C.c_void_p] * 4 + [C.c_int32]
        _lib.llkv_tpch_gen_customer.restype = None
        _lib.llkv_tpch_gen_customer.argtypes = [C.c_uint64, C.c_double, C.c_uint64, C.c_uint64] + [C.c_void_p] * 2 + [C.c_int32]
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def gen_lineitem(rows: int, scale: float, columns: Optional[Sequence[str]] = None, row_begin: int = 0,
                 seed: int = SEED, threads: int = 0) -> Dict[str, np.ndarray]:
    """Columns of ``lineitem`` rows [row_begin, row_begin+rows); flags are 1-byte chars."""
    names = list(LINEITEM_SCHEMA) if columns is None else list(columns)
    out = {n: np.empty(rows, dtype=_NP[LINEITEM_SCHEMA[n][1]]) for n in names}
    order = ["l_orderkey", "l_partkey", "l_suppkey", "l_linenumber", "l_quantity", "l_extendedprice", "l_discount",
             "l_tax", "l_shipdate", "l_commitdate", "l_receiptdate", "l_returnflag", "l_linestatus"]
    gen_lib().llkv_tpch_gen_lineitem(seed, scale, row_begin, rows, *[_ptr(out.get(n)) for n in order], threads)
    return out


# The reference's own TPC-H DDL declares the money columns DECIMAL(15,2) (the dss.ddl its harness installs, llkv-tpch/src/
# lib.rs:154,1027-1091: values parsed into PlanValue::Decimal at the column's scale): the same synthetic rows with those four
# columns as Decimal128(15, 2) raw values — quantity · 100, prices in cents, discount and tax in hundredths.
DECIMAL_MONEY = ("l_quantity", "l_extendedprice", "l_discount", "l_tax")
DECIMAL_PRECISION, DECIMAL_SCALE = 15, 2


def lineitem_as_decimal(data: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """`gen_lineitem` output with the DECIMAL_MONEY columns replaced by their int64 raw values at scale 2 (exact: the generator
    makes every float as cents / 100.0)."""
    out = dict(data)
    for name in DECIMAL_MONEY:
        if name in out:
            out[name] = np.rint(out[name] * 100.0).astype(np.int64) if out[name].dtype != np.int64 else out[name] * 100
    return out


def lineitem_dtype(name: str, decimal: bool = False) -> int:
    return abi.DT_DECIMAL128 if decimal and name in DECIMAL_MONEY else LINEITEM_SCHEMA[name][1]


def gen_orders(rows: int, scale: float, row_begin: int = 0, seed: int = SEED, threads: int = 0) -> Dict[str, np.ndarray]:
    out = {n: np.empty(rows, dtype=_NP[ORDERS_SCHEMA[n][1]]) for n in ORDERS_SCHEMA}
    gen_lib().llkv_tpch_gen_orders(seed, scale, row_begin, rows, _ptr(out["o_orderkey"]), _ptr(out["o_custkey"]),
                                   _ptr(out["o_orderdate"]), _ptr(out["o_shippriority"]), threads)
    return out


def gen_customer(rows: int, scale: float, row_begin: int = 0, seed: int = SEED, threads: int = 0) -> Dict[str, np.ndarray]:
    out = {"c_custkey": np.empty(rows, np.int64), "c_mktsegment": np.empty(rows, np.uint8)}
    gen_lib().llkv_tpch_gen_customer(seed, scale, row_begin, rows, _ptr(out["c_custkey"]), _ptr(out["c_mktsegment"]), threads)
    return out


def orders_for_lineitems(n: int) -> int:
    return int(gen_lib().llkv_tpch_orders_for_lineitems(n))


def customers_for_scale(scale: float) -> int:
    return int(gen_lib().llkv_tpch_customers_for_scale(scale))


def chunk_rows(total_rows: int, chunk: int = CHUNK_ROWS) -> List[int]:
    full, rem = divmod(total_rows, chunk)
    return [chunk] * full + ([rem] if rem else [])


# ------------------------------------------------------------------------- queries
@dataclass
class QueryPlan:
    name: str
    predicate: List[Filter]
    aggs: List[AggregateSpec]
    keys: List[int]
    order_by_keys: bool
    columns: List[str]  # lineitem columns that must be staged
    bytes_per_row: int  # SURVEY.md §8(d) algorithmic bytes

    @property
    def grouped(self) -> bool:
        return bool(self.keys)


def c1() -> QueryPlan:
    """configs[0]: SELECT sum(l_extendedprice) FROM lineitem WHERE l_quantity < 24"""
    return QueryPlan("c1", [Filter(L_QUANTITY, Operator.LessThan(24))], [AggregateSpec.sum(L_EXTENDEDPRICE)], [], False,
                     ["l_quantity", "l_extendedprice"], 16)


def q6() -> QueryPlan:
    """TPC-H Q6: sum(l_extendedprice * l_discount), 3-predicate conjunction (dates as Date32 day numbers)."""
    pred = [
        Filter(L_SHIPDATE, Operator.Range(Bound.Included(DATE_1994_01_01), Bound.Excluded(DATE_1995_01_01))),
        Filter(L_DISCOUNT, Operator.Range(Bound.Included(0.05), Bound.Included(0.07))),
        Filter(L_QUANTITY, Operator.LessThan(24)),
    ]
    return QueryPlan("q6", pred, [AggregateSpec.sum(col(L_EXTENDEDPRICE) * col(L_DISCOUNT), "revenue")], [], False,
                     ["l_shipdate", "l_discount", "l_quantity", "l_extendedprice"], 28)


def q1() -> QueryPlan:
    """TPC-H Q1: GROUP BY l_returnflag, l_linestatus with 8 aggregates, ORDER BY the keys."""
    price, disc, tax = col(L_EXTENDEDPRICE), col(L_DISCOUNT), col(L_TAX)
    aggs = [
        AggregateSpec.sum(L_QUANTITY, "sum_qty"),
        AggregateSpec.sum(L_EXTENDEDPRICE, "sum_base_price"),
        AggregateSpec.sum(price * (1 - disc), "sum_disc_price"),
        AggregateSpec.sum(price * (1 - disc) * (1 + tax), "sum_charge"),
        AggregateSpec.avg(L_QUANTITY, "avg_qty"),
        AggregateSpec.avg(L_EXTENDEDPRICE, "avg_price"),
        AggregateSpec.avg(L_DISCOUNT, "avg_disc"),
        AggregateSpec.count_star("count_order"),
    ]
    return QueryPlan("q1", [Filter(L_SHIPDATE, Operator.LessThanOrEquals(DATE_1998_09_02))], aggs,
                     [L_RETURNFLAG, L_LINESTATUS], True,
                     ["l_shipdate", "l_returnflag", "l_linestatus", "l_quantity", "l_extendedprice", "l_discount", "l_tax"], 38)


QUERIES = {"c1": c1, "q6": q6, "q1": q1}


def lineitem_column_descs(rows: int, keep: list, decimal: bool = False):
    """llkv_column_desc[] for plan lowering without data: statistics and dictionaries as
    staging discovers them on the synthetic data (quantity 1..50; flags in first-appearance
    order N/R/A and O/F).  `decimal`: the DECIMAL(15,2) form of the money columns with the
    statistics of their SF10 raw values."""
    descs = (abi.CColumnDesc * len(LINEITEM_SCHEMA))()
    dicts = {"l_returnflag": [b"N", b"R", b"A"], "l_linestatus": [b"O", b"F"]}
    stats = {"l_quantity": (1, 50), "l_linenumber": (1, 7), "l_shipdate": (8036, 10561)}
    for i, (name, (fid, dt)) in enumerate(LINEITEM_SCHEMA.items()):
        d = descs[i]
        d.field_id, d.dtype, d.rows = fid, dt, rows
        if name in stats:
            d.has_stats, d.min_i, d.max_i = 1, stats[name][0], stats[name][1]
        if decimal and name in DECIMAL_MONEY:
            dstats = {"l_quantity": (100, 5000), "l_extendedprice": (90091, 10494950), "l_discount": (0, 10), "l_tax": (0, 8)}
            d.dtype, d.precision, d.scale = abi.DT_DECIMAL128, DECIMAL_PRECISION, DECIMAL_SCALE
            d.has_stats, d.min_i, d.max_i = 1, dstats[name][0], dstats[name][1]
        if name in dicts:
            arr = (C.c_char_p * len(dicts[name]))(*dicts[name])
            keep.append(arr)
            d.dict_size, d.dictionary = len(dicts[name]), arr
    return descs
