python -m pytest tests/test_gpu_overlap.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for cl in 9 10 11; do
  echo "CAPLOG $cl"
  for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat; do
  HAO_BIN_CAPLOG=$cl python bench.py --workload $wl --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['value']/1e6,2), d['ms_per_step'], 'bins', d['stage_ms'].get('q_sort_bins'), 'ovl', d['config']['overlaps_per_gpu_step'])"
  done
done
