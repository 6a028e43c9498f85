cd $GRAFT_REPO_ROOT; O=gpurun_out/c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_digest.py tests/test_gpu_overlap.py tests/test_gpu_altpaths.py tests/test_gpu_chr1.py -x -q -m gpu -s 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --cpu-baseline none > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ('value','ms_per_step','value_boundary','boundary','roofline','stage_ms'): print(k, d.get(k))
PY
timeout 300 python bench.py --cpu-baseline none --workload bacterial5M_hifi30x > $O/bench_5M.json 2>> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_5M.json").read().strip().splitlines()[-1])
for k in ('value','ms_per_step','value_boundary','boundary','stage_ms'): print(k, d.get(k))
PY
