# round 6: the measured arena placement: delivery tests, the probe forced on configs[2] (what every NUMA node gives), then three default delivered runs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ag; mkdir -p $O
ls /sys/devices/system/node/ | tr '\n' ' '; cat /sys/devices/system/node/online; 
timeout 600 python -m pytest tests/test_gpu_stream.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
HAO_DBG_TEST=arena_probe=1 HAO_DBG_PRINT=dl timeout 600 python bench.py --cpu-baseline none --no-variants --no-verify --steps 3 --warmup 1 > $O/probe.json 2> $O/probe.err; grep -h "arena" $O/probe.err | head -40
for i in 1 2 3; do
timeout 600 python bench.py --cpu-baseline none --no-variants --no-verify --steps 5 > $O/run$i.json 2> $O/run$i.err; grep -h "\[hao\]" $O/run$i.err | head -4
python - $O/run$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']
print('delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', round(b['copy_gb_per_s'],1))
PY
done
