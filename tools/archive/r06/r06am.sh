# round 6: fewer, larger batches per pass
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06am; mkdir -p $O
for spec in "b6:" "b5:--batch-reads 100000" "b4:--batch-reads 125000" "b6b:"; do name=${spec%%:*}; fl=${spec#*:}
timeout 600 python bench.py --cpu-baseline none --no-variants --no-verify --steps 6 --warmup 2 $fl > $O/$name.json 2> $O/$name.err; tail -1 $O/$name.err | cut -c1-200
python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']
    print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', round(b['copy_gb_per_s'],1), 'batches', b['batches_per_pass'], 'mem', d['device_memory']['used_gb_at_end_of_run'], 'seed', round(d['stage_ms']['q_sort_bins'],1), 'chain', round(d['stage_ms']['q_chain'],1))
except Exception as e: print(sys.argv[2], 'no line', e)
PY
done
