# generic A/B on the GPU box: usage bash tools/r04_ab.sh <tag> "<pytest targets or empty>" name:ENV=V,ENV2=V:workload:extra-bench-args ...
tag=$1; tests=$2; shift; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
if [ -n "$tests" ]; then timeout 1200 python -m pytest $tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log; fi
for spec in "$@"; do
  IFS=: read name envs wl extra <<< "$spec"
  envv=(); IFS=, read -ra ea <<< "$envs"; for x in "${ea[@]}"; do [ -n "$x" ] && envv+=("$x"); done
  env "${envv[@]}" X_=1 timeout 900 python bench.py --cpu-baseline none --no-variants --workload $wl $extra > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']; b=(d.get('boundary') or {})
    print(f"{sys.argv[2]:18s} resident {d['ms_per_step_resident']:9.2f} delivered {b.get('ms_per_step',0):9.2f} | seed {s.get('q_sort_bins',0):7.2f} chain {s.get('q_chain',0):7.2f} dp {s.get('q_chain_dp',0):6.2f} asm {s.get('q_assemble',0):6.2f} sel {s.get('q_select',0):7.2f} fin {s.get('q_final',0):5.2f} | pt {s.get('pt_sort',0)+s.get('pt_count',0)+s.get('pt_lookup',0)+s.get('pt_table',0):6.2f} ok {(b.get('delivered_bytes_check') or {}).get('equal_to_reference')} seqgroups {d['config'].get('groups_on_sequential_path')}")
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done
