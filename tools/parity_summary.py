"""Reduce gpurun_out/parity_observed.jsonl (written by tests/gpu_util.record / observe during `pytest -m gpu`) to one JSON:
per test the recorded metrics (maximum over repeated records of the same numeric key) - the file kept as profiles/rNN_parity_observed.json,
against which the tolerances written in the tests are set (about twice the observed value)."""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_observed.jsonl"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_parity_observed.json"
tests = {}
for line in open(src):
    row = json.loads(line)
    name = row.pop("test")
    if "kind" in row:                                  # observe(): one (kind, value, tol) triple per record
        row = {row["kind"]: row["value"], "tol_" + row["kind"]: row["tol"]}
    t = tests.setdefault(name, {})
    for k, v in row.items():
        if isinstance(v, (int, float)) and not k.startswith("tol") and isinstance(t.get(k), (int, float)):
            t[k] = max(t[k], v)
        else:
            t[k] = v
json.dump({"_how": "observed errors recorded by the -m gpu tests (tests/gpu_util.record / observe), maxima per test", "tests": tests},
          open(dst, "w"), indent=1)
print(dst, len(tests), "tests")
