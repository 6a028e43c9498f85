cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ac; mkdir -p $O
for spec in split: nosplit:--no-tail-split; do IFS=: read name fl <<< "$spec"
timeout 600 python bench.py --cpu-baseline none --no-variants --no-verify --steps 5 $fl > $O/$name.json 2> $O/$name.err; echo "rc=$?"
python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=b['stage_ms']
print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy ms', b['copy_ms_per_step'], 'wait', b['host_ms_in_wait'], 'async', b['host_ms_in_async'], 'stage sum', round(sum(s.values()),1), 'resident sum', round(sum(d['stage_ms'].values()),1))
print('   delivered stages', s)
PY
done
