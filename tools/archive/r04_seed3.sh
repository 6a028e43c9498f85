# A/B of the many-bin seed launches: usage bash tools/r04_seed3.sh <tag>
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_stream.py tests/test_gpu_edge.py tests/test_gpu_zz_new.py tests/test_gpu_fullgold.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --cpu-baseline none --no-variants "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']; b=(d.get('boundary') or {})
    print(f"{sys.argv[2]:20s} resident {d['ms_per_step_resident']:9.2f} delivered {b.get('ms_per_step',0):9.2f} seed {s.get('q_sort_bins',0):8.2f} chain {s.get('q_chain',0):7.2f} sel {s.get('q_select',0):7.2f} asm {s.get('q_assemble',0):6.2f} ok {(b.get('delivered_bytes_check') or {}).get('equal_to_reference')}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run rr_direct4 X=1 -- --workload chr1_250M_hifi30x_repeat --steps 2
run rr_direct8 HAO_SEED_NU=8 -- --workload chr1_250M_hifi30x_repeat --steps 2 --no-boundary
run rr_staged HAO_SEED_NODIRECT=1 -- --workload chr1_250M_hifi30x_repeat --steps 2 --no-boundary
run rr5_direct4 X=1 -- --workload bacterial5M_hifi30x_repeat --steps 5
run rr5_staged HAO_SEED_NODIRECT=1 -- --workload bacterial5M_hifi30x_repeat --steps 5
run ont_direct4 X=1 -- --workload ont50M_30x --steps 3 --no-boundary
run ont_staged HAO_SEED_NODIRECT=1 -- --workload ont50M_30x --steps 3 --no-boundary
