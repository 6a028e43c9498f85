cd $GRAFT_REPO_ROOT; O=gpurun_out/c6; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stream.py tests/test_gpu_overlap.py tests/test_gpu_exact.py tests/test_gpu_chr1.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --cpu-baseline none > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ('value','ms_per_step','value_boundary','boundary','roofline','stage_ms'): print(k, d.get(k))
PY
bash tools/r02_counters.sh c6
