# round 6: the seed kernel's reads handed out by a cursor instead of fixed shares: tests, then the resident step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ad; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_altpaths.py tests/test_gpu_fullgold.py -q -m gpu -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --cpu-baseline none --no-variants --no-boundary --steps 20 --warmup 5 > $O/bench$i.json 2> $O/bench$i.err; echo "rc=$?"
python - $O/bench$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']
print('resident', d['ms_per_step_resident'], 'seed', s['q_sort_bins'], 'per launch', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'chain', s['q_chain'])
PY
done
