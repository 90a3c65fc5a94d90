"""Per-dispatch table of one codec decode from a rocprofv3 --kernel-trace CSV (argv[1] = *_kernel_trace.csv, argv[2] = number of
trailing dispatches to print): kernel, grid, duration.  Workload: tools/pmc_codec_probe.py snac32 | q3b32."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
tot = 0
for r in rows[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    g = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    w = "x".join(r.get(k, "?") for k in ("Workgroup_Size_X",))
    print(f"{r['Kernel_Name'][:60]:60s} grid {g:22s} wg {w:5s} {d:10.1f} us")
print("total us", tot)
