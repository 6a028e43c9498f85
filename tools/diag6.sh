timeout 300 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
run() { timeout 90 python bench.py --workload $1 --no-cpu-baseline 2>/tmp/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(d['config']['workload'], round(d['value']/1e6,2), d['ms_per_step'], 'bins', s.get('q_sort_bins'), 'ovl', d['config']['overlaps_per_gpu_step'])"; }
for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x; do run $wl; done
