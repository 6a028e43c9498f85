# round 3: whole GPU suite, then benches with the boundary measurement (variants through the environment: V="name:ENV=1 name2:")
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/seed_t.log 2>&1
tail -14 gpurun_out/seed_t.log
for spec in ${V:-default: noperm:HAO_SEED_NOPERM=1}; do
  v=${spec%%:*}; e=${spec#*:}
  for wl in ${WL:-chr1_250M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x}; do
    env $e timeout 400 python bench.py --workload $wl --no-cpu-baseline --steps 3 --warmup 1 2>gpurun_out/seed_b_${v}_$wl.err | tail -1 > gpurun_out/seed_b_${v}_$wl.json
    python -c "
import json,sys
d=json.loads(open('gpurun_out/seed_b_${v}_$wl.json').read()); print('$v', d['config']['workload'], round(d['value']/1e6,2), d['ms_per_step'], 'boundary', d['value_boundary'], d['boundary']['ms_per_step'], d['stage_ms'], d['boundary']['stage_ms'])"
  done
done
