# round 6: what bounds the list-major seed kernel?  Its DBG instance with the hit stores / the record loads taken out (results wrong on purpose)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O
for f in 0 1 2 3; do HAO_SEED_LDS=1 HAO_DBG_SEEDPHASE=1 HAO_DBG_SEEDFLAGS=$f timeout 600 python bench.py --workload chr1_250M_hifi30x --cpu-baseline none --no-variants --no-boundary --steps 2 --warmup 1 > $O/f$f.json 2> $O/f$f.err; echo "flags $f"; grep "seed lds" $O/f$f.err | tail -2; python - $O/f$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('  seed stage ms', round(d['stage_ms']['q_sort_bins'],2))
PY
done
