# round 6: hand-made arenas on huge pages, probe threshold 50 GB/s: delivery tests, then four default delivered runs with the arena lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06al; mkdir -p $O
grep -h . /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_stream.py -q -m gpu -x > $O/pytest.log 2>&1; tail -1 $O/pytest.log
for i in 1 2 3 4; do
HAO_DBG_PRINT=dl timeout 600 python bench.py --cpu-baseline none --no-variants --no-verify --steps 5 > $O/run$i.json 2> $O/run$i.err; grep -h "\[hao\]\|\] arena" $O/run$i.err | cut -c1-200 | head -6
python - $O/run$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']
print('delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', round(b['copy_gb_per_s'],1), 'wait', b['host_ms_in_wait'])
PY
done
