"""MFMA utilisation and implied clock per kernel from one rocprofv3 pass
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d DIR -- python tools/pmc_codec_probe.py <what>
  python tools/pmc_mfma_reduce.py DIR out.json
mfma_util = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 * 1024 SIMDs): GRBM_GUI_ACTIVE as reported is summed over the 8 XCDs
(the profiles/r01_pmc/mfma_util.json formula); clock_GHz = (GRBM_GUI_ACTIVE / 8) / kernel duration from the trace of the same run."""
import csv, glob, json, os, sys
d, out = sys.argv[1], sys.argv[2]
cnt, dur = {}, {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        e = cnt.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            e["calls"] = e.get("calls", 0) + 1
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        dur[k] = dur.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {"_how": __doc__.strip().split("\n")[1].strip(), "kernels": {}}
for k, e in sorted(cnt.items(), key=lambda kv: -dur.get(kv[0], 0)):
    gui = e.get("GRBM_GUI_ACTIVE", 0.0)
    if not gui or dur.get(k, 0) < 2e5:
        continue
    res["kernels"][k] = {"calls": e.get("calls", 0), "total_ms": round(dur[k] / 1e6, 3), "mfma_util": round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 4),
                         "clock_GHz": round(gui / 8 / dur[k], 3), "cu_busy": round(e.get("SQ_BUSY_CU_CYCLES", 0.0) / (gui / 8 * 256), 3)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=0)[:3000])
