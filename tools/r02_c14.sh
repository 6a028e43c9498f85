cd $GRAFT_REPO_ROOT; O=gpurun_out/c14; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $O/pytest.log; head -2 $O/pytest.log
for wl in chr1_250M_hifi30x bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x; do timeout 600 python bench.py --cpu-baseline none --workload $wl > $O/bench_$wl.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", d['ms_per_step'], round(d['value']/1e6,2), round((d['value_boundary'] or 0)/1e6,2), d['roofline']['frac'], d['stage_ms'])
PY
done
