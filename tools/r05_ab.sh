# round 5: A/B runs on the GPU box.  usage: bash tools/r05_ab.sh <tag> "<pytest targets run first, stop on failure, or empty>" "<pytest targets run LAST or empty>" name:ENV=V,ENV2=V:workload:extra-bench-args ...
tag=$1; tests=$2; tests_last=$3; shift; shift; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
if [ -n "$tests" ]; then timeout 1200 python -m pytest $tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log; fi
for spec in "$@"; do
  IFS=: read name envs wl extra <<< "$spec"
  envv=(); IFS=, read -ra ea <<< "$envs"; for x in "${ea[@]}"; do [ -n "$x" ] && envv+=("$x"); done
  env "${envv[@]}" X_=1 timeout 900 python bench.py --cpu-baseline none --no-variants --workload $wl $extra > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']; b=(d.get('boundary') or {}); rf=d.get('roofline') or {}
    print(f"{sys.argv[2]:14s} resident {d['ms_per_step_resident']:8.2f} delivered {b.get('ms_per_step',0):8.2f} | seed {s.get('q_sort_bins',0):7.2f} chain {s.get('q_chain',0):6.2f} dp {s.get('q_chain_dp',0):5.2f} asm {s.get('q_assemble',0):5.2f} sel {s.get('q_select',0):6.2f} | roofline {rf.get('kernel','')[:28]} {rf.get('achieved',0):7.1f} frac {rf.get('frac',0):.3f} | ok {(b.get('delivered_bytes_check') or {}).get('equal_to_reference')}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
if [ -n "$tests_last" ]; then timeout 1500 python -m pytest $tests_last -x -q -m gpu --durations=5 > $O/pytest_last.log 2>&1; echo "pytest rc=$?" >> $O/pytest_last.log; tail -12 $O/pytest_last.log; fi
