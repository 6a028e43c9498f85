# round 6: ol->list as 32-byte wire records: delivery tests, then the delivered step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ab; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stream.py tests/test_gpu_fullgold.py tests/test_gpu_dropin.py tests/test_gpu_exact.py tests/test_gpu_zz_rankshare.py -q -m gpu -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 python bench.py --cpu-baseline none --no-variants --steps 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?"
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=b['stage_ms']
print('delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'GB/step', b['host_bytes_per_gpu_step']/1e9, 'copy ms', b['copy_ms_per_step'], 'copy GB/s', round(b['copy_gb_per_s'],1), 'wait', b['host_ms_in_wait'], 'ok', b['delivered_bytes_check']['equal_to_reference'])
print('   delivered stages', s)
PY
