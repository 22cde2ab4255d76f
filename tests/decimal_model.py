"""Python-integer model of the reference's exact decimal arithmetic — llkv-types/src/decimal.rs:58-110 (DecimalValue::new:
scale within ±38, at most 38 digits) and llkv-compute/src/scalar/decimal.rs:128-234 (add / sub at the larger scale, mul at the
sum of the scales, div to a target scale with the reference's rounding as written) — used by the tests as an INDEPENDENT check
of the oracle's C restatement (oracle/llkv_oracle.c: dec_binary) and of the GPU path at sizes the oracle does not reach.
Hand-derived from the cited lines: it documents the restatement, it does not pin it (no test of the reference holds these)."""
from __future__ import annotations

MAX_PRECISION = 38


class DecimalError(Exception):
    pass


def digits(v: int) -> int:
    """digit_count_i256: 1 for zero."""
    return len(str(abs(v)))


def new(value: int, scale: int):
    if not -MAX_PRECISION <= scale <= MAX_PRECISION:
        raise DecimalError("scale")
    if digits(value) > MAX_PRECISION:
        raise DecimalError("precision")
    return (value, scale)


def _i128(v: int) -> int:
    if not -(1 << 127) <= v < (1 << 127):
        raise DecimalError("overflow")
    return v


def _i256(v: int) -> int:
    if not -(1 << 255) <= v < (1 << 255):
        raise DecimalError("overflow")
    return v


def _tdiv(a: int, b: int) -> int:
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def rescale_up(d, target: int):
    v, s = d
    if target == s:
        return d
    return new(_i128(_i256(v * 10 ** (target - s))), target)


def binary(l, r, op: str):
    """l, r: (raw, scale); op in + - * /.  None = NULL (a zero divisor, llkv-executor/src/lib.rs:7300-7302)."""
    if op in "+-":
        t = max(l[1], r[1])
        a, b = rescale_up(l, t), rescale_up(r, t)
        return new(_i128(a[0] + b[0] if op == "+" else a[0] - b[0]), t)
    if op == "*":
        s = l[1] + r[1]
        if not -MAX_PRECISION <= s <= MAX_PRECISION:
            raise DecimalError("scale")
        return new(_i128(l[0] * r[0]), s)
    if r[0] == 0:
        return None
    target = l[1]  # "preserve scale of left operand" :7303-7304
    adj = target + r[1] - l[1]
    num = l[0]
    if adj > 0:
        if adj > 2 * MAX_PRECISION:
            raise DecimalError("scale")
        num = _i256(num * 10 ** adj)
    elif adj < 0:
        f = 10 ** (-adj)
        if num - _tdiv(num, f) * f != 0:
            raise DecimalError("inexact")
        num = _tdiv(num, f)
    den = r[0]
    q = _tdiv(num, den)
    rem = num - q * den
    # :209-227 — half = denominator / 2 (truncated), |rem| >= |half| rounds; the direction follows the signs of the TRUNCATED
    # quotient and the denominator, so an odd denominator rounds up from (|d| − 1) / 2 and −1 / 3 comes back as +1
    if rem != 0 and abs(rem) >= abs(_tdiv(den, 2)):
        q = q + 1 if (q >= 0) == (den >= 0) else q - 1
    return new(_i128(q), target)


def temp_column(values):
    """The group's temp column for a computed decimal argument (plan_values_to_arrow_array llkv-executor/src/lib.rs:298-330):
    Decimal128(digits of the first non-NULL value, its scale); arrow refuses a positive scale above the precision."""
    nn = [v for v in values if v is not None]
    if not nn:
        return None
    p, s = digits(nn[0][0]), nn[0][1]
    if s > 0 and s > p:
        raise DecimalError("precision/scale")
    return p, s
