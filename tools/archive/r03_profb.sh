# kernel stats of the boundary (delivery) step: bash tools/r03_profb.sh <workload> <tag>
wl=$1; tag=$2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profb_$tag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --workload $wl --cpu-baseline none --steps 2 --warmup 1 > $O.log 2>&1
cd $R; f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_boundary_$tag.csv && python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f} pct {r['Percentage']}")
PY
rm -rf $O
