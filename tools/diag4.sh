python -m pytest tests/test_gpu_overlap.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
run() { python bench.py --workload $1 --no-cpu-baseline 2>/tmp/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(d['config']['workload'], round(d['value']/1e6,2), d['ms_per_step'], 'groups', s.get('q_groups'), 'chain', s.get('q_chain'), 'dp', s.get('q_chain_dp'), 'ovl', d['config']['overlaps_per_gpu_step'])"; grep "^\[dp\]" /tmp/err.txt | tail -1; }
for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat; do
  echo default; HAO_DBG_DP_STATS=1 run $wl
  echo NOSPEC; HAO_DBG_DP_NOSPEC=1 run $wl
  echo SERIAL; HAO_DBG_DP_SERIAL=1 run $wl
done
