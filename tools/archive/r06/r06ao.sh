# round 6: hao_pack_bits_kernel with four chunks per thread: delivery tests, delivered step, the pack kernels' times
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ao; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_fullgold.py -q -m gpu -x > $O/pytest.log 2>&1; tail -1 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --cpu-baseline none --no-variants --steps 6 --warmup 2 > $O/run$i.json 2> $O/run$i.err
python - $O/run$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=b['stage_ms']
print('delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'sel', s['q_select'], 'asm', s['q_assemble'], 'final', s['q_final'], 'ok', b['delivered_bytes_check']['equal_to_reference'])
PY
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --cpu-baseline none --no-variants --no-verify --steps 2 --warmup 1 > $O/prof.log 2>&1; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i "pack\|rank4\|qtab\|read_ranges" "$f" | cut -c1-60,100-200; rm -rf $O/prof )
