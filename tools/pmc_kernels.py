"""Raw per-kernel counter sums of rocprofv3 --pmc passes -> JSON + table (all counters found, every value divided by SQ_WAVE_CYCLES where that helps reading).
usage: python tools/pmc_kernels.py <dir with one or more passes> [kernel name substring ...]      (default: the 12 kernels with the most wave-cycles)
Counters of several passes (separate directories below <dir>) are merged per kernel name; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles."""
import csv, glob, json, re, sys
root = sys.argv[1]; pats = sys.argv[2:]
acc = {}
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        name = re.sub(r"<.*", "", name) if "rocprim::" in name else name
        a = acc.setdefault(name, {})
        k = r["Counter_Name"]
        a[k] = a.get(k, 0.0) + float(r["Counter_Value"])
        a.setdefault("_disp_" + k, set()).add(r["Dispatch_Id"])
out = {}
for name, a in acc.items():
    d = {k: v for k, v in a.items() if not k.startswith("_disp_")}
    d["dispatches"] = max(len(v) for k, v in a.items() if k.startswith("_disp_"))
    out[name] = d
json.dump(out, open(root.rstrip("/") + ".json", "w"), indent=1)
keys = sorted(out, key=lambda n: -out[n].get("SQ_WAVE_CYCLES", out[n].get("GRBM_GUI_ACTIVE", 0.0)))
sel = [n for n in keys if any(p in n for p in pats)] if pats else keys[:12]
for n in sel:
    d = out[n]; wc = d.get("SQ_WAVE_CYCLES", 0.0)
    print(f"== {n[:90]}  dispatches {d['dispatches']}")
    for k in sorted(d):
        if k == "dispatches": continue
        extra = f"  ({d[k] / wc:.4f} of wave-cycles)" if wc and k.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_LDS", "SQ_INST_LEVEL")) else ""
        print(f"   {k:34s} {d[k]:18.0f}{extra}")
