cd $GRAFT_REPO_ROOT; O=gpurun_out/c5; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --cpu-baseline none > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ('value','ms_per_step','value_boundary','boundary','roofline','stage_ms'): print(k, d.get(k))
PY
timeout 300 python bench.py --cpu-baseline none --workload bacterial5M_hifi30x > $O/bench_5M.json 2>> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_5M.json").read().strip().splitlines()[-1])
for k in ('value','ms_per_step','value_boundary','boundary','stage_ms'): print(k, d.get(k))
PY
