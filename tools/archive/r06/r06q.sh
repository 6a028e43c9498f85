# round 6: the length-jitter fixture through the full-size tests; the default bench line with all its variants (wall clock of the whole command)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullgold.py -q -m gpu -x -k "chr1jit" --durations=5 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
t0=$(date +%s); timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); v=d.get('variants') or {}
print('value', d['value'], 'ms', d['ms_per_step'], 'resident', d['ms_per_step_resident'], 'roofline', {k:d['roofline'][k] for k in ('kernel','achieved','frac','kernel_ms')}, d['roofline'].get('seed_stage'))
print('check', (d.get('boundary') or {}).get('delivered_bytes_check'))
for k,x in v.items():
    print(k, x.get('value'), x.get('ms_per_step'), x.get('ms_per_step_resident'), ((x.get('boundary') or {}).get('delivered_bytes_check') or {}).get('equal_to_reference'), (x.get('roofline') or {}).get('kernel'), (x.get('roofline') or {}).get('frac'), x.get('device_memory'), x.get('prediction_8_gpus'))
print('cpu', d.get('cpu_baseline'))
PY
tail -5 $O/bench_default.err
