# round 6: where the delivery arenas land (HAO_DBG_DLTIME) and what a hand-bound arena (HAO_ARENA_NUMA=4) delivers at, same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stream.py -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for spec in dflt: hand:HAO_ARENA_NUMA=4 plain:HAO_ARENA_NUMA=0; do IFS=: read name envs <<< "$spec"
  env ${envs:-X_=1} HAO_DBG_DLTIME=1 timeout 600 python bench.py --cpu-baseline none --no-variants --steps 3 > $O/$name.json 2> $O/$name.err
  grep "arena" $O/$name.err | head -4
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']
print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', b['copy_gb_per_s'], 'ok', b['delivered_bytes_check']['equal_to_reference'])
PY
done
