"""Inter-kernel gaps of the decode step from a rocprofv3 --kernel-trace CSV: for every consecutive pair of dispatches on the same
queue, gap = Start[i+1] - End[i]; reported per (previous kernel -> next kernel) for the steady decode (pairs seen >= 100 times), with
the kernels' own durations beside them.  Answers "is the time between the kernels or inside them" with timestamps instead of
step_ms - sum(kernel averages).
Usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> out.json"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
files = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((r["Queue_Id"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort(key=lambda r: (r[0], r[1]))


def short(n):
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


pairs = defaultdict(list)
dur = defaultdict(list)
for a, b in zip(rows, rows[1:]):
    if a[0] != b[0]:
        continue
    pairs[(short(a[3]), short(b[3]))].append(b[1] - a[2])
for r in rows:
    dur[short(r[3])].append(r[2] - r[1])
res = {"pairs": [], "kernels": {}}
tot_gap = 0
for k, v in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 100:
        continue
    v = sorted(v)
    med = v[len(v) // 2]
    res["pairs"].append({"prev": k[0], "next": k[1], "n": len(v), "gap_ns_median": med, "gap_ns_mean": sum(v) / len(v), "p10": v[len(v) // 10],
                         "p90": v[9 * len(v) // 10]})
for k, v in dur.items():
    if len(v) >= 100:
        v = sorted(v)
        res["kernels"][k] = {"n": len(v), "ns_median": v[len(v) // 2], "ns_mean": sum(v) / len(v)}
json.dump(res, open(out, "w"), indent=1)
for p in res["pairs"][:14]:
    print(f"{p['prev'][:34]:34s} -> {p['next'][:34]:34s} n={p['n']:7d} gap median {p['gap_ns_median']:6d} ns mean {p['gap_ns_mean']:8.0f}")
