# round 6: timeline of a RESIDENT step: where the device idles between kernels
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06aj; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -- python $R/bench.py --cpu-baseline none --no-variants --no-boundary --steps 2 --warmup 1 > $O/tr.log 2>&1 )
tail -1 $O/tr.log | cut -c1-200
python tools/timeline.py $O/tr $O/timeline_resident.txt | head -60
rm -rf $O/tr
