# round 6: the selection's size classes on side streams: repeat-rich tests, then the repeat-rich and the configs[2] resident steps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ai; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_altpaths.py tests/test_gpu_fullgold.py tests/test_gpu_zz_new.py -q -m gpu -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for spec in "rr:chr1_250M_hifi30x_repeat:" "c2:chr1_250M_hifi30x:" "c2b:chr1_250M_hifi30x:"; do IFS=: read name wl envs <<< "$spec"
env ${envs:-X_=1} timeout 600 python bench.py --workload $wl --cpu-baseline none --no-variants --no-boundary --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err
python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']
print(sys.argv[2], 'resident', d['ms_per_step_resident'], 'seed', round(s['q_sort_bins'],1), 'chain', round(s['q_chain'],1), 'dp', round(s['q_chain_dp'],1), 'asm', round(s['q_assemble'],1), 'sel', round(s['q_select'],1), 'final', round(s['q_final'],1))
PY
done
