# round 6: chain_group_kernel with a plain prefix sum and one vote per predicate: parity subset, resident + delivered step, stage times
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r06ap}; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest ${TESTS:-tests/test_gpu_overlap.py tests/test_gpu_fullgold.py tests/test_gpu_stream.py tests/test_gpu_edge.py} -q -m gpu -x > $O/pytest.log 2>&1; tail -1 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --cpu-baseline none --no-variants --steps 10 --warmup 3 > $O/run$i.json 2> $O/run$i.err
python - $O/run$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=d['stage_ms']
print('delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'seed', s['q_sort_bins'], 'chain', s['q_chain'], 'sk', s['sk_chunks'], 'ptl', s['pt_lookup'], 'frac', d['roofline']['frac'], 'ok', b['delivered_bytes_check']['equal_to_reference'])
print([(k["kernel"],k["kernel_ms"],k["frac"]) for k in d["roofline"].get("kernels",[])], "asm", s["q_assemble"], "sel", s["q_select"], "final", s["q_final"], "b_asm", b["stage_ms"]["q_assemble"], "b_final", b["stage_ms"]["q_final"])
PY
done
if [ -n "$SELDBG" ]; then HAO_DBG_PRINT=sel timeout 300 python bench.py --cpu-baseline none --no-variants --no-boundary --no-verify --steps 1 --warmup 0 > $O/seldbg.json 2> $O/seldbg.err; grep "\[select\]" $O/seldbg.err | head -8; fi
