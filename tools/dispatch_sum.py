"""Sum one decode (from the last occurrence of a start kernel) of a tools/codec_dispatch_trace.py table, grouped by kernel."""
import sys
L = [l for l in open(sys.argv[1]) if " us" in l and not l.startswith("total")]
idx = max(i for i, l in enumerate(L) if sys.argv[2] in l)
rows = L[idx:]
g = {}
for l in rows:
    k = l.split("(")[0].strip()[:30]
    g[k] = g.get(k, 0) + float(l.split()[-2])
print(sys.argv[1].split("/")[-1], round(sum(g.values()) / 1e3, 2), "ms", len(rows), {k: round(v / 1e3, 2) for k, v in sorted(g.items(), key=lambda x: -x[1])[:8]})
