"""Where the waves of a kernel spend their cycles, and what the LDS does meanwhile, from two rocprofv3 passes:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d DIR1 -- <cmd>
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d DIR2 -- <cmd>
  python tools/pmc_wave_states.py DIR1 DIR2 out.json
MI355X_MICROARCH.md: SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three are
disjoint and sum to about SQ_WAVE_CYCLES.  SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles."""
import csv, glob, json, os, sys
dirs, out = sys.argv[1:-1], sys.argv[-1]
cnt, dur = {}, {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            e = cnt.setdefault(k, {})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            dur[k] = dur.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {"_how": "tools/pmc_wave_states.py (two rocprofv3 --pmc passes, see its header)", "kernels": {}}
for k, e in sorted(cnt.items(), key=lambda kv: -dur.get(kv[0], 0)):
    wc = e.get("SQ_WAVE_CYCLES", 0.0)
    if dur.get(k, 0) < 4e5 or not wc:
        continue
    gui = e.get("GRBM_GUI_ACTIVE", 0.0)
    row = {"total_ms_both_passes": round(dur[k] / 1e6, 3),
           "wave_parked_frac": round(e.get("SQ_WAIT_ANY", 0.0) / wc, 3), "issue_stall_frac": round(e.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
           "issuing_frac": round(e.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3), "lds_issue_stall_frac": round(e.get("SQ_WAIT_INST_LDS", 0.0) / wc, 3),
           "lds_issuing_frac": round(e.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 3)}
    if e.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_share_of_lds_cycles"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 3)
    if gui:
        row["mfma_util"] = round(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 4)
        row["lds_active_share_of_cu_time"] = round(e.get("SQ_LDS_IDX_ACTIVE", 0.0) / (gui / 8 * 256), 3)
    res["kernels"][k] = row
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"], indent=0)[:4000])
