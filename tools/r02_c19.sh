cd $GRAFT_REPO_ROOT; O=gpurun_out/c19; mkdir -p $O
for i in 1 2; do timeout 600 python bench.py --cpu-baseline none --steps 2 > $O/bench$i.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench$i.json").read().strip().splitlines()[-1]); b=d['boundary']
print(d['ms_per_step'], d['value'], d['value_boundary'], b['ms_per_step'], b['assemble_and_pack_ms_per_step'], b['copy_ms_per_step'], b['copy_gb_per_s'], b['host_ms_in_async'], b['host_ms_in_wait'])
PY
done
lscpu | grep -i -E "numa|model name|socket" | head; nproc
