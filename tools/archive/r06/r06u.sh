# round 6: LDS layout per rows-per-thread variant (20 592 records for reads of up to 1024 rows): seed tests, then the default bench line with its variants
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullgold.py tests/test_gpu_zz_rankproxy.py -q -m gpu -x > $O/pytest.log 2>&1 && tail -3 $O/pytest.log && \
HAO_SEEDPHASE=1 timeout 900 python bench.py --steps 5 > $O/bench.json 2> $O/bench.err; tail -3 $O/pytest.log; grep -h "seed lds" $O/bench.err | sort | uniq -c | sort -rn | head -8
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']
print('value', d['value'], 'ms', d['ms_per_step'], 'resident', d.get('ms_per_step_resident'), 'roof', d['roofline']['frac'], d['roofline'].get('kernel'), 'delivered', b['ms_per_step'])
print('stages', d.get('stage_ms'))
for k,v in d.get('variants',{}).items():
    print(k, {kk: v[kk] for kk in v if kk in ('ms_per_step','value','stage_ms','seed_ms','seed_path','prediction_8_gpus','ha_ft_gen_s','ha_pt_gen_ms')})
PY
