# round 6: timeline of a delivered step (kernel + memory-copy trace): where the main queue idles
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06v; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -- python $R/bench.py --cpu-baseline none --no-variants --no-verify --steps 1 --warmup 1 > $O/tr.log 2>&1 )
tail -2 $O/tr.log | cut -c1-300
python tools/timeline.py $O/tr $O/timeline_delivered.txt
head -3 $(find $O/tr -name "*memory_copy_trace.csv" | head -1)
rm -rf $O/tr
