"""How tests/golden/*.json were made.

The reference is Rust and cannot be built or run in this environment (no rustc/cargo, no
network), so no fixture could be *generated* by running it.  The files are a
hand transcription, value by value, of the inputs and asserted outputs of the reference's own
known-answer tests; each case's `source` names the test.  This script re-validates ALL SEVEN
files: the JSON parses, every case (in every section) names its source and says whether the
reference holds its expectation, and no case keeps reference source text (Rust syntax) in a
string.
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))

# file → the sections that hold cases
FILES = {
    "table_scan.json": ["cases"], "joins.json": ["cases"], "aggregates.json": ["cases"], "string_predicates.json": ["cases"],
    "string_scans.json": ["cases"], "join_filters.json": ["expression_filters", "cartesian"], "mvcc.json": ["cases", "count_cases", "basic"],
}
RUST = re.compile(r"\bfn \w+\(|\blet mut\b|\bimpl\b|\.unwrap\(\)|=> \{")


def strings_of(x):
    if isinstance(x, str):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from strings_of(v)
    elif isinstance(x, list):
        for v in x:
            yield from strings_of(v)


def validate(fn, sections):
    with open(os.path.join(HERE, fn)) as f:
        doc = json.load(f)
    assert "source" in doc, fn
    n = 0
    for sec in sections:
        assert sec in doc, (fn, sec)
        for case in (doc[sec] if isinstance(doc[sec], list) else [dict(doc[sec], name=sec)]):  # (a section that is ONE case: a dict)
            assert "source" in case or "source" in doc, (fn, case)
            if sec != "basic":
                assert "name" in case, (fn, case)
            assert any(k.startswith("expect") or k == "property" for k in case), (fn, case.get("name"))  # an asserted answer, or the property the reference's test checks
            for s in strings_of(case):
                assert not RUST.search(s), (fn, case.get("name"), "reference source text in a fixture")
            n += 1
    return n


if __name__ == "__main__":
    present = sorted(f for f in os.listdir(HERE) if f.endswith(".json"))
    assert present == sorted(FILES), (present, sorted(FILES))
    for fn, sections in FILES.items():
        print(fn, validate(fn, sections), "cases ok")
