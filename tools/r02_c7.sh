cd $GRAFT_REPO_ROOT; O=gpurun_out/c7; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --cpu-baseline none --steps 2 > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d['ms_per_step'], d['value_boundary'], d['boundary'])
except Exception as e: print("$tag failed", e, open("$O/bench_$tag.err").read()[-500:])
PY
}
run cs4 HAO_COPY_STREAMS=4
run cs1 HAO_COPY_STREAMS=1
run nosdma HAO_COPY_STREAMS=1 HSA_ENABLE_SDMA=0
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -5
