# round 6: tapered last batches for the delivered pass
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ak; mkdir -p $O
for spec in "plain:" "t7:--taper 0.85,0.7,0.45" "t8:--taper 0.8,0.55,0.4,0.25" "t7b:--taper 0.9,0.7,0.4" "plain2:" "t7_2:--taper 0.85,0.7,0.45"; do name=${spec%%:*}; fl=${spec#*:}
timeout 600 python bench.py --cpu-baseline none --no-variants --no-verify --steps 6 --warmup 2 $fl > $O/$name.json 2> $O/$name.err
python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']; s=b['stage_ms']
print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', round(b['copy_gb_per_s'],1), 'wait', b['host_ms_in_wait'], 'stage sum', round(sum(s.values()),1))
PY
done
