# usage: tools/prof.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/ kernel stats csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
