cd $GRAFT_REPO_ROOT; O=gpurun_out/c13; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --cpu-baseline none --steps 2 > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1]); b=d['boundary']
    print("$tag", d['ms_per_step'], d['value_boundary'], b['ms_per_step'], b['copy_ms_per_step'], b['copy_gb_per_s'], b['host_ms_in_async'])
except Exception as e: print("$tag failed", e, open("$O/bench_$tag.err").read()[-500:])
PY
}
run p1cs4 HAO_COPY_STREAMS=4
run p1cs2 HAO_COPY_STREAMS=2
run p1cs1 HAO_COPY_STREAMS=1
run p0cs1 HAO_COPY_STREAMS=1 HAO_STREAM_PRIO=0
