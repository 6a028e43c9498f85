cd $GRAFT_REPO_ROOT; O=gpurun_out/c8; mkdir -p $O
timeout 600 python bench.py --cpu-baseline none --steps 2 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value_boundary']); print(d['stage_ms']); print(d['boundary'])
PY
