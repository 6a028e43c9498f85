cd $GRAFT_REPO_ROOT; O=gpurun_out/c18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_dropin.py tests/test_gpu_indexfile.py tests/test_gpu_exact.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.log; cat $O/pytest.log
timeout 600 python bench.py --cpu-baseline none > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); b=d['boundary']
print(d['ms_per_step'], d['value'], d['value_boundary'], b['ms_per_step'], b['assemble_and_pack_ms_per_step'], b['copy_ms_per_step'])
PY
