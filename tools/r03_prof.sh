# rocprofv3 kernel stats of one bench workload: bash tools/r03_prof.sh <workload> <tag> [bench args]
wl=$1; tag=$2; shift 2; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$tag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --workload $wl --cpu-baseline none --steps 3 --warmup 1 "$@" > $O.log 2>&1
cd $R; f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/kernel_stats_$tag.csv && head -40 $f | cut -c1-200
find $O -name "*kernel_trace.csv" -size +5M -delete
