# round 6: how the runtime carries a device-to-host copy (SDMA rows or blit kernels) and what it costs a streaming kernel beside it
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y; mkdir -p $O; export TMPDIR=/tmp
timeout 120 tools/ubench_copy 1024 > $O/ubench_copy.txt 2>&1; cat $O/ubench_copy.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -- $R/tools/ubench_copy 1024 > $O/tr.log 2>&1 )
python - $O/tr <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True); mf = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
ev = []
for r in csv.DictReader(open(kf[0])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:40] + " q" + r.get("Queue_Id", "?")))
if mf:
    for r in csv.DictReader(open(mf[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")))
ev.sort(); t0 = ev[0][0]
for s, e, n in ev:
    if e - s > 2e5 or n.startswith("C "): print(f"+{(s - t0) / 1e6:9.2f} ms  {(e - s) / 1e6:8.2f} ms  {n}")
PY
rm -rf $O/tr
