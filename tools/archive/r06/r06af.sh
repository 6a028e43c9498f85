# round 6: two batch contexts (host threads, hao_attach) for the pass, resident and delivered
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06af; mkdir -p $O
for spec in "one:" "two12:--contexts 2 --boundary-contexts 2 --batch-reads 42000" "two8:--contexts 2 --boundary-contexts 2 --batch-reads 62500" "two6:--contexts 2 --boundary-contexts 2"; do name=${spec%%:*}; fl=${spec#*:}
timeout 600 python bench.py --cpu-baseline none --no-variants --steps 5 $fl > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; tail -1 $O/$name.err | cut -c1-200
python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=d['boundary']
    print(sys.argv[2], 'delivered', b['ms_per_step'], 'resident', d['ms_per_step_resident'], 'copy GB/s', round(b['copy_gb_per_s'],1), 'ok', (b.get('delivered_bytes_check') or {}).get('equal_to_reference'), 'mem', d.get('device_memory'))
except Exception as e: print(sys.argv[2], 'no line', e)
PY
done
