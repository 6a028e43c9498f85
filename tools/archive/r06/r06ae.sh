# round 6: cursor against fixed shares in the seed kernel, same box, alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ae; mkdir -p $O
for spec in cursor1: static1:HAO_X_STATIC=1 cursor2: static2:HAO_X_STATIC=1 cursor3: static3:HAO_X_STATIC=1; do IFS=: read name envs <<< "$spec"
env ${envs:-X_=1} timeout 600 python bench.py --cpu-baseline none --no-variants --no-boundary --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err
python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']
print(sys.argv[2], 'resident', d['ms_per_step_resident'], 'seed', s['q_sort_bins'], 'per launch', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
PY
done
