timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x; do timeout 120 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['value']/1e6,2), d['ms_per_step'], d['stage_ms'])"; done
