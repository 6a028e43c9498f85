"""Summarise a rocprofv3 --pmc SQ_* pass into the sketch kernel's instruction counts (profiles/r02/sketch_alu.json).
usage: python tools/sketch_alu.py <counter dir> <bases in the pass> [kernel name prefix]"""
import csv, glob, json, re, sys
root, bases = sys.argv[1], float(sys.argv[2]); pref = sys.argv[3] if len(sys.argv) > 3 else "sketch_unit_kernel"
acc, disp_seen = {}, set()
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if not name.startswith(pref): continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        disp_seen.add(r["Dispatch_Id"])
n = max(1, len(disp_seen))
out = {"kernel": pref, "dispatches": n, "bases": bases, "counters_sum": acc}
if "SQ_INSTS_VALU" in acc:
    out["valu_wave_insts_per_base"] = acc["SQ_INSTS_VALU"] / bases
if "SQ_INSTS_SALU" in acc:
    out["salu_wave_insts_per_base"] = acc["SQ_INSTS_SALU"] / bases
if "SQ_INSTS_LDS" in acc:
    out["lds_wave_insts_per_base"] = acc["SQ_INSTS_LDS"] / bases
if "SQ_WAVE_CYCLES" in acc and acc.get("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in acc: out[k.lower() + "_over_wave_cycles"] = acc[k] / acc["SQ_WAVE_CYCLES"]
print(json.dumps(out, indent=1))
