run() { python bench.py --workload $1 --no-cpu-baseline 2>/tmp/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']; print(d['config']['workload'], round(d['value']/1e6,2), d['ms_per_step'], 'chain', s.get('q_chain'), 'dp', s.get('q_chain_dp'), 'ovl', d['config']['overlaps_per_gpu_step'])"; }
for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x; do
  for m in 0 1 2 3; do echo MINCLS $m; HAO_SPEC_MINCLS=$m run $wl; done
done
