# round 6: repeat-rich twin: the list-major kernel for the reads below a hit limit, the table kernels for the rest
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06an; mkdir -p $O
for spec in "tables:" "lds24k:HAO_SEED_LDS_RATIO=1000000" "lds16k:HAO_SEED_LDS_RATIO=1000000 HAO_SEED_MERGE_MAXN=16000" "lds13k:HAO_SEED_LDS_RATIO=1000000 HAO_SEED_MERGE_MAXN=13000" "lds11k:HAO_SEED_LDS_RATIO=1000000 HAO_SEED_MERGE_MAXN=11000" "lds9k:HAO_SEED_LDS_RATIO=1000000 HAO_SEED_MERGE_MAXN=9000"; do name=${spec%%:*}; envs=${spec#*:}
env ${envs:-X_=1} HAO_DBG_PRINT=seed timeout 600 python bench.py --workload chr1_250M_hifi30x_repeat --cpu-baseline none --no-variants --no-boundary --steps 2 --warmup 1 > $O/$name.json 2> $O/$name.err
python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']
print(sys.argv[2], 'resident', d['ms_per_step_resident'], 'seed', round(s['q_sort_bins'],1), d['roofline'].get('seed_stage'))
PY
done
