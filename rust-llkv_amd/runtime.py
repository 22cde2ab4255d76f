"""Harness-side binding of libllkv_hip.so (include/llkv_hip.h) — the same calls a Rust
``extern "C"`` shim inside llkv-executor would make.

There is deliberately no CPU fallback: if the HIP library is missing or no device can be
bound, every data-path call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from . import abi
from .abi import (AggregateSpec, CPlan, CValue, Expr, Filter, LlkvError, Value)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLKV_HIP_LIB") or os.path.join(_HERE, "libllkv_hip.so")  # override: A/B builds of the same ABI

_lib = None


def lib():
    """Loads the C-ABI library.  torch (if importable) is imported first so both share one
    HIP runtime (same SONAME libamdhip64.so.7) and streams / device pointers interoperate."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing — the HIP extension is not built "
                           "(run __graft_entry__.build()); there is no CPU fallback")
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH)
    L.llkv_hip_last_error.restype = C.c_char_p
    L.llkv_hip_abi_version.restype = C.c_uint32
    L.llkv_hip_table_total_rows.restype = C.c_uint64
    L.llkv_hip_table_total_rows.argtypes = [C.c_void_p]
    L.llkv_hip_table_local_rows.restype = C.c_uint64
    L.llkv_hip_table_local_rows.argtypes = [C.c_void_p]
    L.llkv_hip_staging_stats.restype = None
    L.llkv_hip_table_free.argtypes = [C.c_void_p]
    L.llkv_hip_table_free.restype = None
    L.llkv_hip_query_free.argtypes = [C.c_void_p]
    L.llkv_hip_query_free.restype = None
    L.llkv_hip_query_algorithmic_bytes.restype = C.c_uint64
    L.llkv_hip_query_algorithmic_bytes.argtypes = [C.c_void_p]
    L.llkv_hip_query_kernel_signature.restype = C.c_char_p
    L.llkv_hip_query_kernel_signature.argtypes = [C.c_void_p]
    for name in ("llkv_hip_query_num_groups", "llkv_hip_query_num_keys", "llkv_hip_query_num_aggregates"):
        getattr(L, name).restype = C.c_uint32
        getattr(L, name).argtypes = [C.c_void_p]
    L.llkv_hip_free.argtypes = [C.c_void_p]
    L.llkv_hip_free.restype = None
    L.llkv_plan_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise LlkvError(rc, lib().llkv_hip_last_error().decode(errors="replace"))


_bound_device: Optional[int] = None


def init(device: int = 0):
    """llkv_hip_init: bind this process to one GPU (one process per GPU)."""
    global _bound_device
    check(lib().llkv_hip_init(C.c_int32(device)))
    _bound_device = device


def device_count() -> int:
    return int(lib().llkv_hip_device_count())


def staging_stats():
    """(bytes, seconds) of the host → HBM copies this process has made so far."""
    b, t = C.c_uint64(), C.c_double()
    lib().llkv_hip_staging_stats(C.byref(b), C.byref(t))
    return b.value, t.value


def pinned_stats():
    """(cached, outstanding) bytes of the library's page-locked host memory: resting in its block cache / held by results."""
    c, o = C.c_uint64(), C.c_uint64()
    lib().llkv_hip_pinned_stats(C.byref(c), C.byref(o))
    return c.value, o.value


def shutdown():
    global _bound_device
    lib().llkv_hip_shutdown()
    _bound_device = None


# ---------------------------------------------------------------------------------------------------
# Collectives behind the boundary (include/llkv_hip.h "Collectives"): RCCL, or a transport the host supplies
# ---------------------------------------------------------------------------------------------------
COMM_ID_BYTES = 128
_ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_int64), C.c_uint64, C.c_void_p)
_ALL_GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class _CommTransport(C.Structure):
    _fields_ = [("all_reduce_sum_i64", _ALL_REDUCE_FN), ("all_gather", _ALL_GATHER_FN), ("user", C.c_void_p)]


_comm_keep = None  # the ctypes callbacks of a custom transport must outlive the communicator


def comm_unique_id() -> bytes:
    """Rank 0: a fresh ncclUniqueId to hand to the other ranks over the host's own channel."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib().llkv_hip_comm_unique_id(buf))
    return bytes(buf)


def comm_init(unique_id: bytes, rank: int, world: int):
    """Every rank, after init(): join the RCCL communicator (ncclCommInitRank on the bound device)."""
    if len(unique_id) != COMM_ID_BYTES:
        raise ValueError("a unique id has 128 bytes")
    buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
    check(lib().llkv_hip_comm_init(buf, C.c_uint32(rank), C.c_uint32(world)))


def comm_init_torch(dist, rank: int, world: int, group=None):
    """A host-supplied transport over a torch.distributed group whose backend takes CPU tensors (gloo): the library's
    collectives then see host memory (llkv_hip_comm_init_custom) — how the tests run the sharded drivers without RCCL,
    and how a host with its own network layer would plug in."""
    global _comm_keep
    import torch

    def all_reduce(buf, n, _user):
        try:
            t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(int(n),)))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception:  # nothing may unwind across the C boundary
            return 1

    def all_gather(send, recv, nbytes, _user):
        try:
            n = int(nbytes)
            src = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(n,)))
            dst = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(n * world,)))
            dist.all_gather(list(dst.split(n)), src, group=group)
            return 0
        except Exception:
            return 1

    t = _CommTransport(_ALL_REDUCE_FN(all_reduce), _ALL_GATHER_FN(all_gather), None)
    check(lib().llkv_hip_comm_init_custom(C.byref(t), C.c_uint32(rank), C.c_uint32(world)))
    _comm_keep = t


def comm_destroy():
    global _comm_keep
    lib().llkv_hip_comm_destroy()
    _comm_keep = None


def comm_world() -> int:
    f = lib().llkv_hip_comm_world
    f.restype = C.c_uint32
    return int(f())


def comm_describe():
    """(backend, ranks): 'none' / 'rccl' / 'host transport', and the rank count the communicator itself reports (ncclCommCount)."""
    b, r = C.c_int32(), C.c_uint32()
    check(lib().llkv_hip_comm_describe(C.byref(b), C.byref(r)))
    return {0: "none", 1: "rccl", 2: "host transport"}[b.value], r.value


def comm_all_reduce_i64(device_ptr: int, n: int, stream: int = 0):
    """In-place SUM of a device buffer of int64 lanes over the ranks, ordered on ``stream``."""
    check(lib().llkv_hip_comm_all_reduce_i64(C.c_void_p(device_ptr), C.c_uint64(n), C.c_void_p(stream)))


def comm_all_gather_v(payload: bytes) -> List[bytes]:
    """Variable-length all-gather of host bytes: every rank's contribution, in rank order."""
    world = comm_world()
    out, offs = C.c_void_p(), (C.c_uint64 * (world + 1))()
    src = (C.c_uint8 * max(1, len(payload))).from_buffer_copy(payload or b"\0")
    check(lib().llkv_hip_comm_all_gather_v(src, C.c_uint64(len(payload)), C.byref(out), offs))
    try:
        blob = C.string_at(out, offs[world]) if offs[world] else b""
    finally:
        lib().llkv_hip_free(out)
    return [blob[offs[r]:offs[r + 1]] for r in range(world)]


def comm_union_strings(local: Sequence[str]) -> List[str]:
    """Sorted union of the ranks' string lists: the table-wide dictionary of a sharded Utf8 column."""
    enc = [s.encode() for s in local]
    arr = (C.c_char_p * max(1, len(enc)))(*enc)
    out, n = C.POINTER(C.c_char_p)(), C.c_uint32()
    check(lib().llkv_hip_comm_union_strings(arr, C.c_uint32(len(enc)), C.byref(out), C.byref(n)))
    try:
        return [out[i].decode() for i in range(n.value)]
    finally:
        lib().llkv_hip_free(out)


def set_exact_f64_sums(on: bool):
    """llkv_hip_set_exact_f64_sums: queries prepared from now on keep every f64 SUM / AVG / TOTAL exactly."""
    f = lib().llkv_hip_set_exact_f64_sums
    f.restype = None
    f(C.c_int32(1 if on else 0))


def max_threads() -> int:
    """configured_thread_count of the reference's pool (llkv-threading/src/lib.rs:22-31) as the library sees it."""
    f = lib().llkv_hip_max_threads
    f.restype = C.c_uint32
    return int(f())


class SelectShape(C.Structure):
    """llkv_select_shape: the fields of SelectPlan the executor's dispatch looks at."""
    _fields_ = [("has_compound", C.c_int32), ("n_tables", C.c_uint32), ("n_group_by", C.c_uint32), ("n_aggregates", C.c_uint32),
                ("has_computed_aggregates", C.c_int32), ("n_joins", C.c_uint32), ("has_having", C.c_int32), ("has_distinct", C.c_int32),
                ("has_scalar_subqueries", C.c_int32)]


ROUTE_NAMES = {1: "compound", 2: "no_table", 3: "group_by", 4: "cross_product", 5: "aggregates", 6: "computed_aggregates", 7: "projection"}


def select_route(**shape):
    """llkv_hip_select_route: (route name, served by the GPU path?, message) for a plan shape
    (QueryExecutor::execute_select_with_filter, llkv-executor/src/lib.rs:523-563)."""
    s = SelectShape(**shape)
    route = C.c_int32()
    rc = lib().llkv_hip_select_route(C.byref(s), C.byref(route))
    if rc not in (0, 4):
        check(rc)
    return ROUTE_NAMES[route.value], rc == 0, (lib().llkv_hip_last_error().decode(errors="replace") if rc else "")


class _ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class _ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class HipTable:
    """HBM image of a table's column chunks (the `Table` the executor scans)."""

    def __init__(self, table_id: int, chunk_rows: Sequence[int], rank: int = 0, world: int = 1):
        self._h = C.c_void_p()
        self.chunk_rows = [int(r) for r in chunk_rows]
        arr = (C.c_uint64 * max(1, len(self.chunk_rows)))(*self.chunk_rows)
        check(lib().llkv_hip_table_create(C.c_uint16(table_id), arr, C.c_uint32(len(self.chunk_rows)), C.c_uint32(rank),
                                          C.c_uint32(world), C.byref(self._h)))
        first, count = C.c_uint32(), C.c_uint32()
        check(lib().llkv_hip_table_local_chunks(self._h, C.byref(first), C.byref(count)))
        self.first_chunk, self.n_local_chunks = first.value, count.value
        self.rank, self.world = rank, world
        self._keep: list = []
        self._utf8_fields: set = set()     # what append_chunks needs to know about the staged columns
        self._decimal_fields: set = set()

    @property
    def handle(self):
        return self._h

    @property
    def total_rows(self) -> int:
        return int(lib().llkv_hip_table_total_rows(self._h))

    @property
    def local_rows(self) -> int:
        return int(lib().llkv_hip_table_local_rows(self._h))

    @property
    def local_chunk_rows(self) -> List[int]:
        return self.chunk_rows[self.first_chunk:self.first_chunk + self.n_local_chunks]

    def _split(self, values: np.ndarray) -> List[np.ndarray]:
        """Splits a contiguous local array into the local chunks (zero-copy views)."""
        out, off = [], 0
        for r in self.local_chunk_rows:
            out.append(values[off:off + r])
            off += r
        if off != len(values):
            raise ValueError(f"column has {len(values)} rows, local chunks hold {off}")
        return out

    def append_column(self, field_id: int, dtype: int, values: Union[np.ndarray, Sequence[np.ndarray]], valid=None):
        """Stage one fixed-width column (local rows only): pinned host → hipMemcpyAsync → HBM.
        ``valid``: optional boolean array (local rows), False = NULL cell."""
        chunks = self._split(values) if isinstance(values, np.ndarray) else list(values)
        want = np.dtype(abi.NUMPY_OF_DTYPE[dtype])
        chunks = [np.ascontiguousarray(c, dtype=want) for c in chunks]
        ptrs = (C.c_void_p * max(1, len(chunks)))(*[c.ctypes.data for c in chunks])
        check(lib().llkv_hip_table_append_column(self._h, C.c_uint32(field_id), C.c_int32(dtype), ptrs, C.c_uint32(len(chunks))))
        if valid is not None:
            self.set_column_validity(field_id, valid)

    def set_row_ids(self, row_ids: Union[np.ndarray, Sequence[np.ndarray]]):
        """The table's row ids where they are not 0 … n − 1 (local rows, strictly ascending); reported ids follow them."""
        chunks = self._split(row_ids) if isinstance(row_ids, np.ndarray) else list(row_ids)
        chunks = [np.ascontiguousarray(c, dtype=np.uint64) for c in chunks]
        ptrs = (C.c_void_p * max(1, len(chunks)))(*[c.ctypes.data for c in chunks])
        check(lib().llkv_hip_table_set_row_ids(self._h, ptrs, C.c_uint32(len(chunks))))

    def append_decimal128_column(self, field_id: int, precision: int, scale: int, values, valid=None):
        """Stage a Decimal128(precision, scale) column from Python ints (raw values), an int64 array of raw values or an (n, 2) uint64 buffer."""
        buf = values if isinstance(values, np.ndarray) and values.ndim == 2 else \
            abi.i128_buffer_from_i64(values) if isinstance(values, np.ndarray) and values.dtype == np.int64 else abi.i128_buffer(values)
        chunks = [np.ascontiguousarray(c) for c in self._split(buf)]
        ptrs = (C.c_void_p * max(1, len(chunks)))(*[c.ctypes.data for c in chunks])
        check(lib().llkv_hip_table_append_decimal128_column(self._h, C.c_uint32(field_id), C.c_int32(precision), C.c_int32(scale),
                                                            ptrs, C.c_uint32(len(chunks))))
        self._decimal_fields.add(field_id)
        if valid is not None:
            self.set_column_validity(field_id, valid)

    def local_column_stats(self, field_id: int):
        """(min, max) of this rank's rows of an integer column, or None."""
        has, lo, hi = C.c_int32(), C.c_int64(), C.c_int64()
        check(lib().llkv_hip_table_local_column_stats(self._h, C.c_uint32(field_id), C.byref(has), C.byref(lo), C.byref(hi)))
        return (lo.value, hi.value) if has.value else None

    def set_column_stats(self, field_id: int, lo: int, hi: int):
        """Installs the table-wide (min, max) of an integer column (sharded tables; see dist.share_column_stats)."""
        check(lib().llkv_hip_table_set_column_stats(self._h, C.c_uint32(field_id), C.c_int64(lo), C.c_int64(hi)))

    def local_column_float_stats(self, field_id: int):
        """(largest |v|, smallest non-zero |v|) over the finite values of this rank's rows of a float column, or None."""
        has, hi, lo = C.c_int32(), C.c_double(), C.c_double()
        check(lib().llkv_hip_table_local_column_float_stats(self._h, C.c_uint32(field_id), C.byref(has), C.byref(hi), C.byref(lo)))
        return (hi.value, lo.value) if has.value else None

    def set_column_float_stats(self, field_id: int, abs_max: float, abs_min_nonzero: float):
        """Installs the table-wide float statistics of a sharded table's column (what share_metadata agrees on)."""
        check(lib().llkv_hip_table_set_column_float_stats(self._h, C.c_uint32(field_id), C.c_double(abs_max), C.c_double(abs_min_nonzero)))

    def local_column_all_finite(self, field_id: int) -> bool:
        """No NaN / ±∞ among this rank's rows of a float column."""
        v = C.c_int32()
        check(lib().llkv_hip_table_local_column_all_finite(self._h, C.c_uint32(field_id), C.byref(v)))
        return bool(v.value)

    def set_column_all_finite(self, field_id: int, all_finite: bool):
        """Installs the table-wide answer (what share_metadata agrees on)."""
        check(lib().llkv_hip_table_set_column_all_finite(self._h, C.c_uint32(field_id), C.c_int32(1 if all_finite else 0)))

    def share_metadata(self):
        """Sharded tables, before any query is prepared: all ranks agree on integer statistics and on which columns
        have NULL cells (llkv_hip_table_share_metadata over the communicator)."""
        check(lib().llkv_hip_table_share_metadata(self._h))

    def set_column_validity(self, field_id: int, valid):
        """NULL cells of a staged column as one Arrow validity bitmap per local chunk (LSB first)."""
        valid = np.asarray(valid, dtype=bool)
        bitmaps = [np.packbits(c, bitorder="little") if len(c) else np.zeros(1, dtype=np.uint8) for c in self._split(valid)]
        ptrs = (C.c_void_p * max(1, len(bitmaps)))(*[b.ctypes.data for b in bitmaps])
        check(lib().llkv_hip_table_set_column_validity(self._h, C.c_uint32(field_id), ptrs, C.c_uint32(len(bitmaps))))

    def append_utf8_column(self, field_id: int, strings: Union[np.ndarray, Sequence], dictionary: Optional[Sequence[str]] = None, valid=None):
        """Stage a Utf8 column.  ``strings`` is either a uint8 array of 1-byte strings (Arrow
        offsets are then 0..n) or a sequence of Python strings.  ``dictionary`` fixes the codes
        (required for sharded tables: every rank must pass the same table-wide dictionary)."""
        chunks_off, chunks_data = [], []
        if isinstance(strings, np.ndarray) and strings.dtype == np.uint8:
            ramp = np.arange(max(self.local_chunk_rows, default=0) + 1, dtype=np.int32)  # offsets 0..n of any chunk
            for c in self._split(strings):
                chunks_off.append(ramp)
                chunks_data.append(np.ascontiguousarray(c))
        else:
            off = 0
            strings = list(strings)
            if valid is None and any(x is None for x in strings):
                valid = [x is not None for x in strings]
            strings = ["" if x is None else x for x in strings]
            for r in self.local_chunk_rows:
                enc = [s.encode() for s in strings[off:off + r]]
                off += r
                lens = np.fromiter((len(e) for e in enc), dtype=np.int64, count=len(enc))
                offsets = np.zeros(len(enc) + 1, dtype=np.int32)
                np.cumsum(lens, out=offsets[1:])
                chunks_off.append(offsets)
                chunks_data.append(np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy())
        poff = (C.c_void_p * max(1, len(chunks_off)))(*[c.ctypes.data for c in chunks_off])
        pdat = (C.c_void_p * max(1, len(chunks_data)))(*[c.ctypes.data for c in chunks_data])
        if dictionary is None:
            dptr, dn = None, 0
        else:
            enc = [d.encode() for d in dictionary]
            dptr, dn = (C.c_char_p * max(1, len(enc)))(*enc), len(enc)
        check(lib().llkv_hip_table_append_utf8_column(self._h, C.c_uint32(field_id), poff, pdat, C.c_uint32(len(chunks_off)), dptr, C.c_uint32(dn)))
        self._utf8_fields.add(field_id)
        if valid is not None:
            self.set_column_validity(field_id, valid)

    @property
    def generation(self) -> int:
        f = lib().llkv_hip_table_generation
        f.restype = C.c_uint64
        f.argtypes = [C.c_void_p]
        return int(f(self._h))

    def key_images(self):
        """llkv_hip_table_key_images: (images held, their device bytes)."""
        n, b = C.c_uint32(), C.c_uint64()
        f = lib().llkv_hip_table_key_images
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        check(f(self._h, C.byref(n), C.byref(b)))
        return int(n.value), int(b.value)

    def append_chunks(self, chunk_rows: Sequence[int], columns: Dict[int, object], valid: Optional[Dict[int, object]] = None, row_ids=None):
        """llkv_hip_table_append_chunks: ``chunk_rows`` new chunks behind the table's last one.  ``columns``: field id → the NEW rows of
        that column (numpy array of the staged dtype; int64 raw values / Python ints / (n, 2) uint64 for Decimal128; a uint8 array of
        1-byte strings or a sequence of str / None for Utf8) — every staged column must be there.  ``valid``: field id → boolean array
        (False = NULL cell).  ``row_ids``: uint64 ids of the new rows (a table with its own ids needs them).  Queries prepared before
        the call must be prepared again."""
        chunk_rows = [int(r) for r in chunk_rows]
        n_new, total = len(chunk_rows), sum(chunk_rows)
        keep = []

        def split(a):
            out, off = [], 0
            for r in chunk_rows:
                out.append(a[off:off + r])
                off += r
            if off != len(a):
                raise ValueError(f"{len(a)} new rows given, the new chunks hold {off}")
            return out

        def ptrs(arrs):
            arrs = [np.ascontiguousarray(a) for a in arrs]
            keep.append(arrs)
            p = (C.c_void_p * max(1, len(arrs)))(*[a.ctypes.data if a.size else None for a in arrs])
            keep.append(p)
            return C.cast(p, C.POINTER(C.c_void_p))
        valid = dict(valid or {})
        cc = (abi.CColumnChunks * max(1, len(columns)))()
        for i, (fid, vals) in enumerate(columns.items()):
            cc[i].field_id = fid
            if fid in self._utf8_fields:
                if isinstance(vals, np.ndarray) and vals.dtype == np.uint8:
                    offs = [np.arange(r + 1, dtype=np.int32) for r in chunk_rows]
                    dats = [np.ascontiguousarray(c) if len(c) else np.zeros(1, np.uint8) for c in split(vals)]
                else:
                    vals = list(vals)
                    if fid not in valid and any(x is None for x in vals):
                        valid[fid] = [x is not None for x in vals]
                    offs, dats = [], []
                    for c in split(["" if x is None else x for x in vals]):
                        enc = [x.encode() for x in c]
                        o = np.zeros(len(enc) + 1, dtype=np.int32)
                        np.cumsum([len(e) for e in enc], out=o[1:])
                        offs.append(o)
                        dats.append(np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy())
                cc[i].offsets, cc[i].data = ptrs(offs), ptrs(dats)
            elif fid in self._decimal_fields:
                buf = vals if isinstance(vals, np.ndarray) and vals.ndim == 2 else \
                    abi.i128_buffer_from_i64(vals) if isinstance(vals, np.ndarray) and vals.dtype == np.int64 else abi.i128_buffer(vals)
                cc[i].values = ptrs(split(buf))
            else:
                cc[i].values = ptrs(split(np.asarray(vals)))
            if fid in valid:
                v = np.asarray(valid[fid], dtype=bool)
                cc[i].validity = ptrs([np.packbits(c, bitorder="little") if len(c) else np.zeros(1, np.uint8) for c in split(v)])
        rows_arr = (C.c_uint64 * max(1, n_new))(*chunk_rows)
        idp = None
        if row_ids is not None:
            idp = ptrs(split(np.ascontiguousarray(row_ids, dtype=np.uint64)))
        check(lib().llkv_hip_table_append_chunks(self._h, rows_arr, C.c_uint32(n_new), cc, C.c_uint32(len(columns)), idp))
        self.chunk_rows += chunk_rows
        self.n_local_chunks += n_new
        del total

    def append_arrow_column(self, field_id: int, chunks, dictionary: Optional[Sequence[str]] = None):
        """Stage a column from pyarrow arrays, one per local chunk (llkv_hip_table_append_arrow_column through the
        Arrow C Data Interface: values, offsets and validity bitmaps are read where they lie)."""
        import pyarrow as pa
        arrs = [(_ArrowArray(), _ArrowSchema()) for _ in chunks]
        for c, (a, s) in zip(chunks, arrs):
            c._export_to_c(C.addressof(a), C.addressof(s))
        try:
            ptrs = (C.POINTER(_ArrowArray) * max(1, len(arrs)))(*[C.pointer(a) for a, _ in arrs])
            if dictionary is None:
                dptr, dn = None, 0
            else:
                enc = [d.encode() for d in dictionary]
                dptr, dn = (C.c_char_p * max(1, len(enc)))(*enc), len(enc)
            schema = C.byref(arrs[0][1]) if arrs else None
            check(lib().llkv_hip_table_append_arrow_column(self._h, C.c_uint32(field_id), schema, ptrs, C.c_uint32(len(arrs)), dptr, C.c_uint32(dn)))
        finally:
            for a, s in arrs:  # hand the exported structs back: pyarrow releases them
                pa.Array._import_from_c(C.addressof(a), C.addressof(s))

    def append_arr0_column(self, field_id: int, blobs: Sequence[bytes], dictionary: Optional[Sequence[str]] = None):
        """Stage a column from its llkv-column-map `ARR0` chunk blobs (one per local chunk)."""
        bufs = [np.frombuffer(b, dtype=np.uint8) for b in blobs]
        ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
        lens = (C.c_uint64 * max(1, len(bufs)))(*[len(b) for b in bufs])
        if dictionary is None:
            dptr, dn = None, 0
        else:
            enc = [d.encode() for d in dictionary]
            dptr, dn = (C.c_char_p * max(1, len(enc)))(*enc), len(enc)
        check(lib().llkv_hip_table_append_arr0_column(self._h, C.c_uint32(field_id), ptrs, lens, C.c_uint32(len(bufs)), dptr, C.c_uint32(dn)))

    def adopt_device_column(self, field_id: int, dtype: int, device_ptr: int):
        check(lib().llkv_hip_table_adopt_device_column(self._h, C.c_uint32(field_id), C.c_int32(dtype), C.c_void_p(device_ptr)))

    def close(self):
        if self._h:
            lib().llkv_hip_table_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupRow:
    def __init__(self, keys: List[Value], values: List[Value]):
        self.keys, self.values = keys, values

    def __repr__(self):
        return f"GroupRow(keys={[k.value for k in self.keys]}, values={[v.value for v in self.values]})"


class PreparedQuery:
    """A lowered plan bound to a table: launch() enqueues the kernels, finish() returns rows."""

    def __init__(self, table: HipTable, predicate, aggs: Sequence[AggregateSpec], keys: Sequence[int] = (),
                 order_by_keys: bool = False):
        self._plan = CPlan(predicate, aggs, keys)
        self._h = C.c_void_p()
        self.table = table
        self.n_aggs = len(aggs)
        p = self._plan
        if keys:
            check(lib().llkv_hip_query_prepare_groupby(table.handle, p.filters, p.n_filters, p.ops, p.n_ops, p.keys, p.n_keys,
                                                       p.aggs, p.n_aggs, C.c_int32(int(order_by_keys)), C.byref(self._h)))
        else:
            check(lib().llkv_hip_query_prepare_aggregate(table.handle, p.filters, p.n_filters, p.ops, p.n_ops, p.aggs, p.n_aggs,
                                                         C.byref(self._h)))

    @property
    def kernel_signature(self) -> str:
        return lib().llkv_hip_query_kernel_signature(self._h).decode()

    @property
    def route_note(self) -> str:
        f = lib().llkv_hip_query_route_note
        f.restype = C.c_char_p
        return f(self._h).decode()

    @property
    def algorithmic_bytes(self) -> int:
        return int(lib().llkv_hip_query_algorithmic_bytes(self._h))

    def launch(self, stream: int = 0):
        check(lib().llkv_hip_query_launch(self._h, C.c_void_p(stream)))

    def exchange_buffer(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        check(lib().llkv_hip_query_exchange_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def set_depth(self, depth: int):
        check(lib().llkv_hip_query_set_depth(self._h, C.c_uint32(depth)))

    def wait_folded(self, stream: int):
        check(lib().llkv_hip_query_wait_folded(self._h, C.c_void_p(stream)))

    def all_reduce(self, stream: int = 0):
        """The query's one collective, inside the library (RCCL): the exchange image of the oldest execution not yet
        submitted is summed over the ranks on ``stream``."""
        check(lib().llkv_hip_query_all_reduce(self._h, C.c_void_p(stream)))

    def finish_sharded(self, stream: int = 0) -> List["GroupRow"]:
        """finish() over a sharded table, collectives included (dense, sort-based and DISTINCT forms)."""
        check(lib().llkv_hip_query_finish_sharded(self._h, C.c_void_p(stream)))
        return self.rows()

    def submit(self, stream: int = 0):
        check(lib().llkv_hip_query_submit(self._h, C.c_void_p(stream)))

    def collect(self) -> List[GroupRow]:
        check(lib().llkv_hip_query_collect(self._h))
        return self.rows()

    def collect_only(self):
        """llkv_hip_query_collect — the library waits for the execution, folds and FINALIZES its groups — without building Python
        rows from them (rows() reads the latest collected execution on request)."""
        check(lib().llkv_hip_query_collect(self._h))

    def finish(self, stream: int = 0) -> List[GroupRow]:
        check(lib().llkv_hip_query_finish(self._h, C.c_void_p(stream)))
        return self.rows()

    def read_exchange(self, stream: int = 0) -> np.ndarray:
        """Copy of this rank's exchange image [8][lanes] (uint64 lanes) — test / host-collective helper."""
        _, n = self.exchange_buffer()
        out = np.zeros(int(n), dtype=np.uint64)
        check(lib().llkv_hip_query_read_exchange(self._h, out.ctypes.data_as(C.c_void_p), C.c_uint64(n)))
        return out.reshape(8, -1)

    def finish_from_host(self, exchange: np.ndarray) -> List[GroupRow]:
        ex = np.ascontiguousarray(exchange, dtype=np.uint64).reshape(-1)
        check(lib().llkv_hip_query_finish_from_host(self._h, ex.ctypes.data_as(C.c_void_p), C.c_uint64(ex.size)))
        return self.rows()

    def lane_ops(self) -> List[int]:
        n = C.c_uint32()
        check(lib().llkv_hip_query_lane_ops(self._h, None, C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        check(lib().llkv_hip_query_lane_ops(self._h, buf, C.byref(n)))
        return list(buf)

    def rows(self) -> List[GroupRow]:
        L = lib()
        out = []
        ng, nk, na = L.llkv_hip_query_num_groups(self._h), L.llkv_hip_query_num_keys(self._h), L.llkv_hip_query_num_aggregates(self._h)
        v = CValue()
        for g in range(ng):
            keys, vals = [], []
            for k in range(nk):
                check(L.llkv_hip_query_group_key(self._h, g, k, C.byref(v)))
                keys.append(Value.from_c(v))
            for a in range(na):
                check(L.llkv_hip_query_value(self._h, g, a, C.byref(v)))
                vals.append(Value.from_c(v))
            out.append(GroupRow(keys, vals))
        return out

    def run(self, stream: int = 0) -> List[GroupRow]:
        self.launch(stream)
        return self.finish(stream)

    def finish_only(self, stream: int = 0):
        """llkv_hip_query_finish without building Python rows (the groups stay in the library's arrays)."""
        check(lib().llkv_hip_query_finish(self._h, C.c_void_p(stream)))

    def distinct_partial(self, agg: int) -> np.ndarray:
        """DISTINCT aggregate ``agg`` over a sharded table: this rank's distinct values (uint64 images) in order of
        first appearance."""
        vals, n = C.POINTER(C.c_uint64)(), C.c_uint64()
        check(lib().llkv_hip_query_distinct_partial(self._h, C.c_uint32(agg), C.byref(vals), C.byref(n)))
        return np.ctypeslib.as_array(vals, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)

    def merge_distinct(self, agg: int, parts):
        """Installs the table-wide value of DISTINCT aggregate ``agg`` from every rank's distinct_partial (rank order)."""
        keep = [np.ascontiguousarray(p, np.uint64) for p in parts]
        counts = (C.c_uint64 * len(keep))(*[len(p) for p in keep])
        ptrs = (C.c_void_p * len(keep))(*[p.ctypes.data for p in keep])
        check(lib().llkv_hip_query_merge_distinct(self._h, C.c_uint32(agg), C.c_uint32(len(keep)), counts, ptrs))

    def partial_groups(self):
        """Sort-based GROUP BY over a sharded table, after launch + finish on this rank: the partial groups of its
        rows as (key_values[n_keys][n] int64, key_valid[n_keys][n] uint8, lanes[n][k] uint64) numpy copies."""
        n, nk, k = C.c_uint64(), C.c_uint32(), C.c_uint32()
        kv, kva, ln = C.POINTER(C.c_int64)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)()
        check(lib().llkv_hip_query_partial_groups(self._h, C.byref(n), C.byref(nk), C.byref(k), C.byref(kv), C.byref(kva), C.byref(ln)))
        if n.value == 0:
            return np.zeros((nk.value, 0), np.int64), np.zeros((nk.value, 0), np.uint8), np.zeros((0, k.value), np.uint64)
        return (np.ctypeslib.as_array(kv, shape=(nk.value, n.value)).copy(), np.ctypeslib.as_array(kva, shape=(nk.value, n.value)).copy(),
                np.ctypeslib.as_array(ln, shape=(n.value, k.value)).copy())

    def merge_groups(self, parts):
        """Installs the table-wide groups from every rank's partial_groups() (in rank order); rows() then reads them."""
        world = len(parts)
        keep = [(np.ascontiguousarray(a, np.int64), np.ascontiguousarray(b, np.uint8), np.ascontiguousarray(c, np.uint64)) for a, b, c in parts]
        counts = (C.c_uint64 * world)(*[c.shape[0] for _, _, c in keep])
        kv = (C.c_void_p * world)(*[a.ctypes.data for a, _, _ in keep])
        kva = (C.c_void_p * world)(*[b.ctypes.data for _, b, _ in keep])
        ln = (C.c_void_p * world)(*[c.ctypes.data for _, _, c in keep])
        check(lib().llkv_hip_query_merge_groups(self._h, C.c_uint32(world), counts, kv, kva, ln))

    def set_profiling(self, enabled):
        """False/0 off, True/1 every launch, n > 1: HIP events around every n-th scan."""
        check(lib().llkv_hip_query_set_profiling(self._h, C.c_int32(int(enabled))))

    def kernel_time(self):
        ms, n, name = C.c_double(), C.c_uint64(), C.c_char_p()
        check(lib().llkv_hip_query_kernel_time(self._h, C.byref(ms), C.byref(n), C.byref(name)))
        return ms.value, n.value, (name.value or b"").decode()

    def close(self):
        if self._h:
            lib().llkv_hip_query_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def aggregate(table: HipTable, predicate, aggs: Sequence[AggregateSpec]) -> List[Value]:
    """execute_aggregates / compute_aggregate_values (llkv-executor/src/lib.rs:5357,6087)."""
    q = PreparedQuery(table, predicate, aggs)
    try:
        return q.run()[0].values
    finally:
        q.close()


def groupby(table: HipTable, predicate, keys: Sequence[int], aggs: Sequence[AggregateSpec], order_by_keys: bool = False) -> List[GroupRow]:
    """execute_group_by_single_table (llkv-executor/src/lib.rs:4405)."""
    q = PreparedQuery(table, predicate, aggs, keys, order_by_keys)
    try:
        return q.run()
    finally:
        q.close()


PRIM_TYPE_CODES = {abi.DT_UINT64: 1, abi.DT_INT32: 2, abi.DT_UINT32: 3, abi.DT_FLOAT32: 4, abi.DT_INT64: 6, abi.DT_FLOAT64: 11,
                   abi.DT_UTF8: 12, abi.DT_DATE32: 16}


def arr0_serialize(dtype: int, values, precision: int = 0, scale: int = 0) -> bytes:
    """Writer of the `ARR0` chunk blob for the types on this path (layout documented in
    llkv-column-map/src/serialization.rs:41-140) — harness / test helper."""
    import struct
    if dtype == abi.DT_BOOLEAN:  # arrow's bit-packed values buffer
        bits = np.packbits(np.asarray(values, dtype=bool), bitorder="little").tobytes()
        return b"ARR0" + bytes([0, 15, 0, 0]) + struct.pack("<QII", len(values), len(bits), 0) + bits
    if dtype == abi.DT_DECIMAL128:  # 16-byte little-endian raw values; precision / scale in the two reserved header bytes
        raw = abi.i128_buffer(values).tobytes()
        return b"ARR0" + bytes([0, 18, precision & 0xFF, scale & 0xFF]) + struct.pack("<QII", len(values), len(raw), 0) + raw
    if dtype == abi.DT_UTF8:
        enc = [s.encode() for s in values]
        offsets = np.zeros(len(enc) + 1, dtype=np.int32)
        np.cumsum([len(e) for e in enc], out=offsets[1:])
        data = b"".join(enc)
        return b"ARR0" + bytes([2, 12, 0, 0]) + struct.pack("<QII", len(enc), offsets.nbytes, len(data)) + offsets.tobytes() + data
    arr = np.ascontiguousarray(values, dtype=np.dtype(abi.NUMPY_OF_DTYPE[dtype]))
    return b"ARR0" + bytes([0, PRIM_TYPE_CODES[dtype], 0, 0]) + struct.pack("<QII", len(arr), arr.nbytes, 0) + arr.tobytes()


def arr0_describe(blob: bytes):
    d = abi.CArr0Desc()
    buf = np.frombuffer(blob, dtype=np.uint8)
    check(lib().llkv_hip_arr0_describe(buf.ctypes.data_as(C.c_void_p), C.c_uint64(len(blob)), C.byref(d)))
    return d


def dense_row_runs(chunks: Sequence) -> tuple:
    """dense_row_runs (llkv-column-map/src/store/scan/filter.rs:1510-1582) over (row_count, min, max) triples."""
    arr = (abi.CChunkMeta * max(1, len(chunks)))()
    for i, (n, lo, hi) in enumerate(chunks):
        arr[i].row_count, arr[i].min_val_u64, arr[i].max_val_u64 = n, lo, hi
    dense, first = C.c_int32(), C.c_uint64()
    check(lib().llkv_hip_dense_row_runs(arr, C.c_uint32(len(chunks)), C.byref(dense), C.byref(first)))
    return bool(dense.value), first.value


def filter_row_ids(table: HipTable, predicate, count_only: bool = False):
    """StorageTable::filter_row_ids (llkv-executor/src/types/storage.rs:34-37): ascending row ids
    (``count_only``: just how many — the ids still reach host memory, but are not copied into numpy)."""
    p = CPlan(predicate)
    out, n = C.POINTER(C.c_uint64)(), C.c_uint64()
    check(lib().llkv_hip_filter_row_ids(table.handle, p.filters, p.n_filters, p.ops, p.n_ops, C.byref(out), C.byref(n)))
    if count_only:
        lib().llkv_hip_free(out)
        return n.value
    res = np.ctypeslib.as_array(out, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)
    lib().llkv_hip_free(out)
    return res


def batch_to_arrow(batch_view, names: Optional[Sequence[str]] = None):
    """A raw ``llkv_batch_view`` (inside a scan callback) as a pyarrow RecordBatch that outlives the callback:
    llkv_hip_batch_export_arrow + the Arrow C Data Interface."""
    import pyarrow as pa
    arr, sch = _ArrowArray(), _ArrowSchema()
    cn = (C.c_char_p * len(names))(*[n.encode() for n in names]) if names else None
    check(lib().llkv_hip_batch_export_arrow(C.byref(batch_view), cn, C.byref(arr), C.byref(sch)))
    return pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))


def decode_view_column(c, n: int) -> list:
    """One ``llkv_column_view`` of n rows as a list of Python values (None = NULL cell)."""
    if c.dtype == abi.DT_DECIMAL128:
        raw = np.frombuffer(C.string_at(c.values, n * 16), dtype=np.uint64).reshape(n, 2)
        vals = [abi.i128_from_words(int(lo), int(np.int64(hi))) for lo, hi in raw]
    elif c.dtype == abi.DT_UTF8:
        codes = np.frombuffer(C.string_at(c.values, n), dtype=np.uint8)
        vals = [c.dictionary[int(k)].decode() for k in codes] if not c.validity else None
        if vals is None:  # a NULL cell's code may be any byte
            bits = np.unpackbits(np.frombuffer(C.string_at(c.validity, (n + 7) // 8), dtype=np.uint8), bitorder="little")[:n]
            return [c.dictionary[int(k)].decode() if ok else None for k, ok in zip(codes, bits)]
    else:
        npdt = np.dtype(abi.NUMPY_OF_DTYPE[c.dtype])
        vals = np.frombuffer(C.string_at(c.values, n * npdt.itemsize), dtype=npdt).tolist()
    if c.validity:  # Arrow validity bitmap → None for NULL cells
        bits = np.unpackbits(np.frombuffer(C.string_at(c.validity, (n + 7) // 8), dtype=np.uint8), bitorder="little")[:n]
        vals = [v if ok else None for v, ok in zip(vals, bits)]
    return vals


def scan_stream(table: HipTable, projections, predicate, include_nulls: bool = False, include_row_ids: bool = False, order=None, consume=None):
    """StorageTable::scan_stream: returns the list of batches [(columns, row_ids)], each column a list of
    Python values.  ``projections``: field ids (ScanProjection::Column) or ScalarExpr (::Computed).
    ``consume``: called with every raw ``llkv_batch_view`` instead (buffers valid during the call only); nothing
    is converted or returned."""
    keep: list = []
    projs = (abi.CProjection * max(1, len(projections)))()
    for i, pr in enumerate(projections):
        if isinstance(pr, int):
            projs[i].computed, projs[i].field_id = 0, pr
        else:
            arr = pr.to_c(keep)
            projs[i].computed, projs[i].expr, projs[i].expr_len = 1, arr, len(pr.tokens)
    p = CPlan(predicate)
    opts = abi.scan_options(include_nulls, include_row_ids, order)
    batches = []

    def on_batch(bp, _user):
        b = bp.contents
        if consume is not None:
            consume(b)
            return
        n = int(b.num_rows)
        cols = [decode_view_column(b.columns[ci], n) for ci in range(b.num_columns)]
        rids = np.frombuffer(C.string_at(b.row_ids, n * 8), dtype=np.uint64).tolist() if b.row_ids else None
        batches.append((cols, rids))

    cb = abi.ON_BATCH(on_batch)
    check(lib().llkv_hip_scan_stream(table.handle, projs, C.c_uint32(len(projections)), p.filters, p.n_filters, p.ops, p.n_ops,
                                     C.byref(opts), cb, None))
    return batches


def join_stream(left: HipTable, right: HipTable, keys, join_type: int = abi.JOIN_INNER, batch_size: int = 8192, consume=None, key_rules: int = 0):
    """TableJoinExt::join_stream (llkv-join/src/lib.rs:240-282): list of batches (left_rows, right_rows|None);
    a right row of 2**64-1 is the NULL padding of a LEFT join.  ``consume(n_pairs)``: called per batch instead
    (nothing is converted or returned)."""
    ck = (abi.CJoinKey * max(1, len(keys)))()
    for i, k in enumerate(keys):
        ck[i].left_field, ck[i].right_field = k[0], k[1]
        ck[i].null_equals_null = int(k[2]) if len(k) > 2 else 0
    opts = abi.CJoinOptions(join_type, batch_size, key_rules)
    batches = []

    def on_batch(pl, pr, n, _u):
        if consume is not None:
            consume(n)
            return
        l = np.ctypeslib.as_array(pl, shape=(n,)).tolist()
        r = np.ctypeslib.as_array(pr, shape=(n,)).tolist() if pr else None
        batches.append((l, r))

    cb = abi.ON_JOIN_BATCH(on_batch)
    check(lib().llkv_hip_join_stream(left.handle, right.handle, ck, C.c_uint32(len(keys)), C.byref(opts), cb, None))
    return batches


def _join_keys(keys):
    ck = (abi.CJoinKey * max(1, len(keys)))()
    for i, k in enumerate(keys):
        ck[i].left_field, ck[i].right_field = k[0], k[1]
        ck[i].null_equals_null = int(k[2]) if len(k) > 2 else 0
    return ck


def join_stream_batches(left: HipTable, right: HipTable, keys, left_columns, right_columns, join_type: int = abi.JOIN_INNER,
                        batch_size: int = 8192, key_rules: int = 0, consume=None):
    """TableJoinExt::join_stream delivering the joined RecordBatches (llkv_hip_join_stream_batches):
    ``left_columns`` / ``right_columns`` = [(field_id, name)] — the user columns of the two schemas.  Returns
    [(names, columns)], each column a list of Python values (None = NULL).  ``consume(batch_view, names)``: called
    with every raw view instead (buffers valid during the call only)."""
    out, keep = abi.join_output(left_columns, right_columns)
    opts = abi.CJoinOptions(join_type, batch_size, key_rules)
    batches = []

    def on_batch(bp, names, _u):
        b = bp.contents
        nm = [names[i].decode() for i in range(b.num_columns)]
        if consume is not None:
            consume(b, nm)
            return
        assert not b.row_ids
        n = int(b.num_rows)
        batches.append((nm, [decode_view_column(b.columns[ci], n) for ci in range(b.num_columns)]))

    cb = abi.ON_JOIN_RECORD_BATCH(on_batch)
    check(lib().llkv_hip_join_stream_batches(left.handle, right.handle, _join_keys(keys), C.c_uint32(len(keys)), C.byref(opts), C.byref(out), cb, None))
    del keep
    return batches


def join_output_names(left_columns, right_columns, join_type: int = abi.JOIN_INNER, key_rules: int = 0) -> List[str]:
    """build_output_schema's names (llkv_hip_join_output_names; host only)."""
    out, keep = abi.join_output(left_columns, right_columns)
    names = (C.c_void_p * (len(left_columns) + len(right_columns) + 1))()
    n = C.c_uint32()
    check(lib().llkv_hip_join_output_names(C.byref(out), C.c_int32(join_type), C.c_int32(key_rules), names, C.byref(n)))
    res = []
    for i in range(n.value):
        res.append(C.string_at(names[i]).decode())
        lib().llkv_hip_free(C.c_void_p(names[i]))
    del keep
    return res


class JoinTopk:
    """The arguments of llkv_hip_join_groupby_topk, marshalled once (what a C caller holds anyway): ``run()`` is the call itself —
    building the filter / token structures from Python objects costs about as much host time as the device spends on a tenth of Q3."""

    def __init__(self, fact: HipTable, fact_filters, fact_key: int, dim: HipTable, dim_filters, dim_key: int, sum_expr,
                 payload_fields: Sequence[int] = (), limit: int = 10, dim_fk: int = 0, dim2: Optional[HipTable] = None,
                 dim2_filters=(), dim2_key: int = 0):
        keep = [fact, dim, dim2]

        def side(table, filters, key):
            p = CPlan(list(filters or []))
            keep.append(p)
            s = abi.CJoinSide()
            s.table, s.filters, s.n_filters, s.key_field = table.handle, p.filters, p.n_filters, key
            return s

        self._f, self._d = side(fact, fact_filters, fact_key), side(dim, dim_filters, dim_key)
        self._d2 = side(dim2, dim2_filters, dim2_key) if dim2 is not None else None
        self._toks, self._n_toks = sum_expr.to_c(keep), len(sum_expr.tokens)
        self._n_payload = len(payload_fields)
        self._pay = (C.c_uint32 * max(1, self._n_payload))(*payload_fields)
        self._limit, self._dim_fk = limit, dim_fk
        self._rows = (abi.CJoinGroupRow * max(1, limit))()
        self._keep = keep
        self._fn = lib().llkv_hip_join_groupby_topk

    def run(self):
        n, total = C.c_uint32(), C.c_uint64()
        check(self._fn(C.byref(self._f), C.byref(self._d), C.c_uint32(self._dim_fk), C.byref(self._d2) if self._d2 is not None else None, self._pay,
                       C.c_uint32(self._n_payload), self._toks, C.c_uint32(self._n_toks), C.c_uint32(self._limit), self._rows, C.byref(n), C.byref(total)))
        out = [(r.key, r.sum, r.count) + tuple(r.payload[i] for i in range(self._n_payload)) for r in self._rows[:n.value]]
        return out, total.value


def join_groupby_topk(fact: HipTable, fact_filters, fact_key: int, dim: HipTable, dim_filters, dim_key: int, sum_expr,
                      payload_fields: Sequence[int] = (), limit: int = 10, dim_fk: int = 0, dim2: Optional[HipTable] = None,
                      dim2_filters=(), dim2_key: int = 0):
    """fact ⋈ dim [⋉ dim2] GROUP BY dim key (+payload) SUM(expr) ORDER BY sum DESC, payload[0] LIMIT k
    (TPC-H Q3 shape; llkv_hip_join_groupby_topk).  Returns (rows, total_groups); rows = (key, sum, count, payload...)."""
    return JoinTopk(fact, fact_filters, fact_key, dim, dim_filters, dim_key, sum_expr, payload_fields, limit, dim_fk, dim2, dim2_filters, dim2_key).run()


class JoinRow:
    """One result row of a join → GROUP BY: the group key (dim.key), the payload cells (None = NULL), the aggregates' finalized
    cells and the dimension row's position among the qualifying rows (the last tie-break of the order)."""

    def __init__(self, key: int, payload: list, values: List[Value], group_index: int):
        self.key, self.payload, self.values, self.group_index = key, payload, values, group_index

    def __repr__(self):
        return f"JoinRow(key={self.key}, payload={self.payload}, values={[v.value for v in self.values]})"


class JoinGroupBy(PreparedQuery):
    """fact ⋈ dim [⋉ dim2] GROUP BY dim key [, payload …] with ANY aggregate list over fact-side expressions
    (llkv_hip_join_groupby_prepare): a prepared GROUP BY of the fact key over `filters AND key IN (qualifying dimension keys)` —
    launch / finish (or finish_sharded for a fact table sharded over ranks) like any other prepared query, then ``result``."""

    def __init__(self, fact: HipTable, fact_filters, fact_key: int, dim: HipTable, dim_filters, dim_key: int, aggs: Sequence[AggregateSpec],
                 dim_fk: int = 0, dim2: Optional[HipTable] = None, dim2_filters=(), dim2_key: int = 0):
        keep = []

        def side(table, filters, key):
            p = CPlan(list(filters or []))
            keep.append(p)
            s = abi.CJoinSide()
            s.table, s.filters, s.n_filters, s.key_field = table.handle, p.filters, p.n_filters, key
            return s

        f, d = side(fact, fact_filters, fact_key), side(dim, dim_filters, dim_key)
        d2 = side(dim2, dim2_filters, dim2_key) if dim2 is not None else None
        self._plan = CPlan(None, aggs, [])
        self._keep = keep
        self._h = C.c_void_p()
        self.table = fact
        self.n_aggs = len(aggs)
        check(lib().llkv_hip_join_groupby_prepare(C.byref(f), C.byref(d), C.c_uint32(dim_fk), C.byref(d2) if d2 is not None else None,
                                                  self._plan.aggs, self._plan.n_aggs, C.byref(self._h)))

    def result(self, payload_fields: Sequence[int] = (), order=(), limit: Optional[int] = None):
        """ORDER BY … LIMIT over the finished groups.  ``order``: [(kind, index, descending, nulls_first)] with kind one of
        abi.JOIN_ORDER_AGGREGATE / _PAYLOAD / _KEY.  Returns (rows, total_groups)."""
        L = lib()
        pay = (C.c_uint32 * max(1, len(payload_fields)))(*payload_fields)
        ok = (abi.CJoinOrderKey * max(1, len(order)))()
        for i, o in enumerate(order):
            ok[i].kind, ok[i].index, ok[i].descending, ok[i].nulls_first = o[0], o[1], int(o[2]) if len(o) > 2 else 0, int(o[3]) if len(o) > 3 else 0
        h = C.c_void_p()
        check(L.llkv_hip_join_groupby_rows(self._h, pay, C.c_uint32(len(payload_fields)), ok, C.c_uint32(len(order)),
                                           C.c_uint64(2**64 - 1 if limit is None else limit), C.byref(h)))
        try:
            L.llkv_hip_join_rows_len.restype = C.c_uint64
            L.llkv_hip_join_rows_total_groups.restype = C.c_uint64
            L.llkv_hip_join_rows_len.argtypes = [C.c_void_p]
            L.llkv_hip_join_rows_total_groups.argtypes = [C.c_void_p]
            n, total = int(L.llkv_hip_join_rows_len(h)), int(L.llkv_hip_join_rows_total_groups(h))
            rows = []
            key, gi = C.c_int64(), C.c_uint64()
            pv, pn = (C.c_int64 * 4)(), (C.c_uint8 * 4)()
            vals = C.POINTER(abi.CValue)()
            for i in range(n):
                check(L.llkv_hip_join_rows_get(h, C.c_uint64(i), C.byref(key), pv, pn, C.byref(gi), C.byref(vals)))
                rows.append(JoinRow(key.value, [None if pn[c] else pv[c] for c in range(len(payload_fields))],
                                    [Value.from_c(vals[a]) for a in range(self.n_aggs)], gi.value))
            return rows, total
        finally:
            L.llkv_hip_join_rows_free(h)


class JoinAgg:
    """The join → GROUP BY → top-k pipeline in phases, for a fact table sharded over ranks
    (llkv_hip_join_agg_*; dist.join_groupby_topk drives the collectives between the phases)."""

    def __init__(self, fact: HipTable, fact_filters, fact_key: int, dim: HipTable, dim_filters, dim_key: int, sum_expr,
                 payload_fields: Sequence[int] = (), dim_fk: int = 0, dim2: Optional[HipTable] = None, dim2_filters=(), dim2_key: int = 0,
                 ranged: bool = False):
        """``ranged``: the range form (llkv_hip_join_agg_prepare_ranged) — raises LlkvError "Unsupported" when the shape
        does not qualify."""
        keep = []

        def side(table, filters, key):
            p = CPlan(list(filters or []))
            keep.append(p)
            s = abi.CJoinSide()
            s.table, s.filters, s.n_filters, s.key_field = table.handle, p.filters, p.n_filters, key
            return s

        f, d = side(fact, fact_filters, fact_key), side(dim, dim_filters, dim_key)
        d2 = side(dim2, dim2_filters, dim2_key) if dim2 is not None else None
        toks = sum_expr.to_c(keep)
        pay = (C.c_uint32 * max(1, len(payload_fields)))(*payload_fields)
        self.n_payload = len(payload_fields)
        self._tables = (fact, dim, dim2)  # the handle reads the tables' HBM images until it is freed: keep them alive
        self._h = C.c_void_p()
        self.ranged = ranged
        prepare = lib().llkv_hip_join_agg_prepare_ranged if ranged else lib().llkv_hip_join_agg_prepare
        check(prepare(C.byref(f), C.byref(d), C.c_uint32(dim_fk), C.byref(d2) if d2 is not None else None, pay,
                      C.c_uint32(len(payload_fields)), toks, C.c_uint32(len(sum_expr.tokens)), C.byref(self._h)))

    def boundary(self) -> bytes:
        """Range form: this rank's block of boundary runs for the all-gather."""
        blk, n = C.c_void_p(), C.c_uint64()
        check(lib().llkv_hip_join_agg_boundary(self._h, C.byref(blk), C.byref(n)))
        return C.string_at(blk, n.value)

    def finish_ranged(self, blocks: Sequence[bytes], rank: int, limit: int):
        """Range form: every rank's boundary block (rank order) → (this rank's candidate rows, groups it reports)."""
        offs = np.zeros(len(blocks) + 1, dtype=np.uint64)
        np.cumsum([len(b) for b in blocks], out=offs[1:])
        buf = np.frombuffer(b"".join(blocks), dtype=np.uint64).copy()
        rows = (abi.CJoinGroupRow * max(1, limit))()
        n, total = C.c_uint32(), C.c_uint64()
        check(lib().llkv_hip_join_agg_finish_ranged(self._h, buf.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint32(len(blocks)),
                                                    C.c_uint32(rank), C.c_uint32(limit), rows, C.byref(n), C.byref(total)))
        return [join_row_tuple(r) for r in rows[:n.value]], total.value

    def exchange_bytes(self) -> int:
        lib().llkv_hip_join_agg_exchange_bytes.restype = C.c_uint64
        return int(lib().llkv_hip_join_agg_exchange_bytes(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().llkv_hip_join_agg_free(self._h)
            self._h = None

    def finish_sharded(self, limit: int):
        """Steps 2–6 inside the library over its communicator: (rows, total_groups), identical on every rank."""
        rows = (abi.CJoinGroupRow * max(1, limit))()
        n, total = C.c_uint32(), C.c_uint64()
        check(lib().llkv_hip_join_agg_finish_sharded(self._h, C.c_uint32(limit), rows, C.byref(n), C.byref(total)))
        return [(r.key, r.sum, r.count) + tuple(r.payload[i] for i in range(self.n_payload)) for r in rows[:n.value]], total.value

    def counts_buffer(self):
        """(device pointer, length) of the int64 per-group row counts to all-reduce (SUM) in place."""
        ptr, n = C.c_void_p(), C.c_uint64()
        check(lib().llkv_hip_join_agg_counts_buffer(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value or 0, n.value

    def straddlers(self):
        """This rank's (groups uint32, values float64) pairs of the groups other ranks hold rows of too."""
        g, v, n = C.POINTER(C.c_uint32)(), C.POINTER(C.c_double)(), C.c_uint64()
        check(lib().llkv_hip_join_agg_straddlers(self._h, C.byref(g), C.byref(v), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.float64)
        return np.ctypeslib.as_array(g, shape=(n.value,)).copy(), np.ctypeslib.as_array(v, shape=(n.value,)).copy()

    def candidates(self, folded, rank: int, limit: int):
        """folded = fold_straddlers(...) → (rows of this rank's report as CJoinGroupRow list, groups reported)."""
        fg, fs, fc, fr = folded
        rows = (abi.CJoinGroupRow * max(1, limit))()
        n, total = C.c_uint32(), C.c_uint64()
        ptr = lambda a, t: a.ctypes.data_as(C.POINTER(t)) if len(a) else None
        check(lib().llkv_hip_join_agg_candidates(self._h, ptr(fg, C.c_uint32), ptr(fs, C.c_double), ptr(fc, C.c_uint64), ptr(fr, C.c_uint32),
                                                 C.c_uint64(len(fg)), C.c_uint32(rank), C.c_uint32(limit), rows, C.byref(n), C.byref(total)))
        return [join_row_tuple(r) for r in rows[:n.value]], total.value


def join_row_tuple(r):
    """CJoinGroupRow → (key, sum, count, payload[4], group_index): a picklable form for all_gather_object."""
    return (int(r.key), float(r.sum), int(r.count), tuple(int(r.payload[i]) for i in range(4)), int(r.group_index))


def fold_straddlers(groups_by_rank, values_by_rank):
    """Host only (llkv_hip_join_agg_fold_straddlers): exact sums of the straddling groups in global row order."""
    world = len(groups_by_rank)
    groups = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.uint32) for g in groups_by_rank]) if world else np.zeros(0, np.uint32))
    values = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in values_by_rank]) if world else np.zeros(0, np.float64))
    offs = np.zeros(world + 1, dtype=np.uint64)
    np.cumsum([len(g) for g in groups_by_rank], out=offs[1:])
    cap = max(1, len(groups))
    og, os_, oc, of = np.zeros(cap, np.uint32), np.zeros(cap, np.float64), np.zeros(cap, np.uint64), np.zeros(cap, np.uint32)
    n = C.c_uint64(cap)
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    check(lib().llkv_hip_join_agg_fold_straddlers(p(groups, C.c_uint32), p(values, C.c_double), p(offs, C.c_uint64), C.c_uint32(world),
                                                  p(og, C.c_uint32), p(os_, C.c_double), p(oc, C.c_uint64), p(of, C.c_uint32), C.byref(n)))
    k = n.value
    return og[:k].copy(), os_[:k].copy(), oc[:k].copy(), of[:k].copy()


def merge_join_rows(rows, n_payload: int, limit: int):
    """Host only (llkv_hip_join_agg_merge): ORDER BY sum DESC, payload[0] ASC, dim row order; LIMIT."""
    arr = (abi.CJoinGroupRow * max(1, len(rows)))()
    for i, t in enumerate(rows):
        arr[i].key, arr[i].sum, arr[i].count, arr[i].group_index = t[0], t[1], t[2], t[4]
        for k in range(4):
            arr[i].payload[k] = t[3][k]
    out = (abi.CJoinGroupRow * max(1, limit))()
    n = C.c_uint32()
    check(lib().llkv_hip_join_agg_merge(arr, C.c_uint32(len(rows)), C.c_uint32(n_payload), C.c_uint32(limit), out, C.byref(n)))
    return [(r.key, r.sum, r.count) + tuple(r.payload[i] for i in range(n_payload)) for r in out[:n.value]]


def lower_plan(column_descs, predicate, aggs: Sequence[AggregateSpec], keys: Sequence[int] = (), grouped: bool = False,
               plan_lib=None, order_by_keys: bool = False, form: int = 0):
    """llkv_plan_lower: returns (type_string, lanes, bytes_per_row) or raises LlkvError.
    form: 0 = per-thread accumulators, 4 = the shared-image lowering, 12 = its partitioned form."""
    L = plan_lib or lib()
    L.llkv_plan_last_error.restype = C.c_char_p
    p = CPlan(predicate, aggs, keys)
    buf = C.create_string_buffer(16384)
    lanes, bpr = C.c_uint32(), C.c_uint64()
    rc = L.llkv_plan_lower(column_descs, C.c_uint32(len(column_descs)), p.filters, p.n_filters, p.ops, p.n_ops, p.keys, p.n_keys,
                           p.aggs, p.n_aggs, C.c_int32(int(grouped) | (2 if (grouped and order_by_keys) else 0) | form), buf, C.c_uint64(len(buf)), C.byref(lanes), C.byref(bpr))
    if rc != 0:
        raise LlkvError(rc, L.llkv_plan_last_error().decode(errors="replace"))
    return buf.value.decode(), lanes.value, bpr.value
