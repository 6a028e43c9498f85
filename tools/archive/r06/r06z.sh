# round 6: the delivered step on the image's HIP runtime (7.2) against the copy bundled with the torch wheel (7.0)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06z; mkdir -p $O; export TMPDIR=/tmp
for spec in bundled: system:HAO_BENCH_SYSTEM_HIP=1; do IFS=: read name envs <<< "$spec"
  env ${envs:-X_=1} timeout 600 python bench.py --cpu-baseline none --no-variants --steps 5 > $O/$name.json 2> $O/$name.err; echo "rc=$?"; tail -2 $O/$name.err | cut -c1-300
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=b['stage_ms']
print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', round(b['copy_gb_per_s'],1), 'seed', s['q_sort_bins'], 'chain', s['q_chain'], 'sel', s['q_select'], 'final', s['q_final'], 'ok', b['delivered_bytes_check']['equal_to_reference'])
print('   resident stages', {k: round(v,1) for k,v in d['stage_ms'].items()})
PY
done
( cd /tmp && HAO_BENCH_SYSTEM_HIP=1 timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -- python $R/bench.py --cpu-baseline none --no-variants --no-verify --steps 1 --warmup 1 > $O/tr.log 2>&1 )
python tools/timeline.py $O/tr $O/timeline_delivered_system_hip.txt | head -12
rm -rf $O/tr
