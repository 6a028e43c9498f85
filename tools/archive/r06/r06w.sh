# round 6: which engine carries the delivery copies (rocclr log), and the delivered step under the runtime's copy switches
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06w; mkdir -p $O; export TMPDIR=/tmp
env | grep -i "sdma\|HSA_\|GPU_\|ROC_\|HIP_" > $O/env.txt
AMD_LOG_LEVEL=4 AMD_LOG_MASK=0x300 timeout 300 python bench.py --workload bacterial5M_hifi30x --cpu-baseline none --no-variants --no-verify --steps 1 --warmup 0 > $O/log_bench.json 2> $O/log.err
grep -c "HSA Copy" $O/log.err; grep -c -i "blit" $O/log.err; grep -i "HSA Copy\|blit\|staging\|pinned" $O/log.err | sed 's/0x[0-9a-f]*/X/g' | cut -c1-220 | sort | uniq -c | sort -rn | head -30 > $O/log_summary.txt; cat $O/log_summary.txt; rm -f $O/log.err
for spec in default: blit0:GPU_FORCE_BLIT_COPY_SIZE=0 wg16:DEBUG_CLR_LIMIT_BLIT_WG=16 wg4:DEBUG_CLR_LIMIT_BLIT_WG=4 sdma_rec0:HSA_ENABLE_SDMA_RECOMMENDED_ENG=0; do IFS=: read name envs <<< "$spec"
  env ${envs:-X_=1} timeout 600 python bench.py --cpu-baseline none --no-variants --no-verify --steps 5 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=b['stage_ms']
print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', round(b['copy_gb_per_s'],1), 'seed', s['q_sort_bins'], 'chain', s['q_chain'], 'sel', s['q_select'], 'final', s['q_final'])
PY
done
