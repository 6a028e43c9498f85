cd $GRAFT_REPO_ROOT; O=gpurun_out/c10; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
HAO_DBG_DLTIME=1 timeout 600 python bench.py --cpu-baseline none > $O/bench.json 2> $O/bench.err; grep deliver $O/bench.err | tail -1; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['value_boundary']); b=d['boundary']; print({k:v for k,v in b.items() if k not in ('what',)})
PY
