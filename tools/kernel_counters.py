"""Per-kernel SQ counter summary of one rocprofv3 --pmc pass (tools/archive/r02_counters.sh) -> JSON + table.
usage: python tools/kernel_counters.py <counter dir> [n_simd=1024]
valu_util = SQ_INSTS_VALU x 2 cycles (a wave64 VALU instruction issues over 2 cycles on a SIMD-32) / (kernel cycles x SIMDs); kernel cycles =
GRBM_GUI_ACTIVE summed over the 8 XCD instances / 8.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (MI355X_MICROARCH.md)."""
import csv, glob, json, re, sys
root = sys.argv[1]; n_simd = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
acc = {}
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        name = re.sub(r"<.*", "", name) if "rocprim::" in name else name
        a = acc.setdefault(name, {"disp": set(), "c": {}})
        a["disp"].add(r["Dispatch_Id"]); a["c"][r["Counter_Name"]] = a["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
out = {}
for name, a in acc.items():
    c = a["c"]; n = len(a["disp"]); d = {"dispatches": n}
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    d["kernel_cycles_total"] = cyc
    if "SQ_INSTS_VALU" in c and cyc: d["valu_util"] = c["SQ_INSTS_VALU"] * 2 / (cyc * n_simd)
    if "SQ_INSTS_SALU" in c and cyc: d["salu_per_simd_cycle"] = c["SQ_INSTS_SALU"] / (cyc * n_simd)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"):
            if k in c: d[k.lower() + "_frac"] = c[k] / wc
        if cyc: d["avg_waves_per_simd"] = wc * 4 / (cyc * n_simd)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"):
        if k in c: d[k] = c[k]
    out[name] = d
json.dump(out, open(root.rstrip("/") + ".json", "w"), indent=1)
rows = sorted(out.items(), key=lambda kv: -kv[1]["kernel_cycles_total"])[:18]
for name, d in rows:
    print(f"{name[:44]:44s} n={d['dispatches']:4d} cyc={d['kernel_cycles_total']/1e6:8.2f}M valu={d.get('valu_util', 0):.2f} waves/simd={d.get('avg_waves_per_simd', 0):.1f} "
          f"wait={d.get('sq_wait_any_frac', 0):.2f} stall={d.get('sq_wait_inst_any_frac', 0):.2f} active={d.get('sq_active_inst_any_frac', 0):.2f}")
