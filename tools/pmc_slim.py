"""keep one row per (kernel, dispatch) of a rocprofv3 counter_collection.csv (sum over instances) - small enough to commit under profiles/"""
import csv, re, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
    name = re.sub(r"<.*", "", name) if "rocprim::" in name else name
    k = (r["Dispatch_Id"], name, r["Counter_Name"])
    acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
w = csv.writer(sys.stdout); w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
for (d, n, c), v in sorted(acc.items(), key=lambda kv: int(kv[0][0])): w.writerow([d, n, c, v])
