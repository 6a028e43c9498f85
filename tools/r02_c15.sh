cd $GRAFT_REPO_ROOT; O=gpurun_out/c15; mkdir -p $O
for pad in 0 6000 14000 27000 54000; do HAO_SEED_LDS_PAD=$pad timeout 600 python bench.py --cpu-baseline none --no-boundary --steps 2 > $O/bench_$pad.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_$pad.json").read().strip().splitlines()[-1])
print("pad $pad", d['ms_per_step'], d['stage_ms']['q_sort_bins'])
PY
done
