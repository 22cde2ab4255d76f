"""How tests/golden/*.json were made.

The reference is Rust and cannot be built or run in this environment (no rustc/cargo, no
network), so no fixture could be *generated* by running it.  The files are a
hand transcription, value by value, of the inputs and asserted outputs of the reference's own
known-answer tests; each case's `source` names the test.  This script only re-validates that
the JSON parses and that every case carries a source citation.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    for fn in ("table_scan.json", "joins.json", "aggregates.json", "string_predicates.json"):
        with open(os.path.join(HERE, fn)) as f:
            doc = json.load(f)
        assert "source" in doc
        for case in doc["cases"]:
            assert "source" in case and "name" in case, case
        print(fn, len(doc["cases"]), "cases ok")
