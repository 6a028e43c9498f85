mkdir -p gpurun_out/r03e; O=gpurun_out/r03e
timeout 400 python -m pytest tests/test_gpu_stream.py tests/test_gpu_dropin.py tests/test_gpu_attach.py tests/test_gpu_exact.py tests/test_gpu_altpaths.py -x -q -m gpu > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-300
for wl in ${WL:-chr1_250M_hifi30x bacterial5M_hifi30x_repeat bacterial5M_hifi30x}; do
  timeout 300 python bench.py --workload $wl --cpu-baseline none --steps 3 --warmup 1 2>$O/bench_$wl.err | tail -1 > $O/bench_$wl.json
  python -c "
import json
d=json.loads(open('$O/bench_$wl.json').read()); print('$wl', d['value'], d['ms_per_step'], d['ms_per_step_resident'], d['boundary']['stage_ms'], d['boundary']['host_bytes_per_gpu_step'], d['boundary']['verbatim_hits_per_step'])"
done
