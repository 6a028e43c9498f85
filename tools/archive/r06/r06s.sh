# round 6: pack kernels on a side stream under the selection: delivery tests, then the delivered step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_fullgold.py -q -m gpu -x -k "stream or wire or chr1-" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for spec in side: serial:HAO_DBG_DP_SERIAL=1; do IFS=: read name envs <<< "$spec"
  env ${envs:-X_=1} timeout 600 python bench.py --cpu-baseline none --no-variants --steps 5 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=b['stage_ms']
print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', b['copy_gb_per_s'], 'asm', s['q_assemble'], 'sel', s['q_select'], 'final', s['q_final'], 'ok', b['delivered_bytes_check']['equal_to_reference'])
PY
done
