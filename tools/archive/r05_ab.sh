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
# optional kernel trace of the default configuration: PROF_WL=<workload> bash tools/r05_ab.sh ...
if [ -n "$PROF_WL" ]; then ( cd /tmp && export TMPDIR=/tmp && env ${PROF_ENV:-X_=1} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --workload $PROF_WL --cpu-baseline none --no-boundary --no-variants --steps 2 --warmup 1 > $O/prof.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -14 $O/kernel_stats.csv | cut -c1-160; rm -rf $O/prof ); fi
if [ -n "$PMC_WL" ]; then ( cd /tmp && export TMPDIR=/tmp && env ${PROF_ENV:-X_=1} timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc -- python $R/bench.py --workload $PMC_WL --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/pmc.log 2>&1
  f=$(find $O/pmc -name "*counter_collection.csv" | head -1); python - "$f" > $O/pmc_seed.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for row in csv.DictReader(open(sys.argv[1])):
    k=row['Kernel_Name'].split('(')[0][:60]
    if not any(x in k for x in ('seed_','chain_group','sketch_unit')): continue
    acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
    key=(k,row['Dispatch_Id'])
    if key not in seen: seen.add(key); n[k]+=1
for k,v in acc.items(): print(k, 'launches', n[k], {c: round(x/n[k]/1e6,2) for c,x in v.items()}, '(per launch, millions; FETCH/WRITE_SIZE in KB units -> x1024 bytes)')
PY
  cat $O/pmc_seed.txt; rm -rf $O/pmc ); fi
