cd $GRAFT_REPO_ROOT; O=gpurun_out/c9; mkdir -p $O
HAO_DBG_DLTIME=1 timeout 600 python bench.py --cpu-baseline none --steps 2 > $O/bench.json 2> $O/bench.err; grep deliver $O/bench.err | tail -4; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value_boundary']); b=d['boundary']; print(b['ms_per_step'], b['host_ms_in_async'])
PY
