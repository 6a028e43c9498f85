# Round-3 measurement run (on the GPU box).  usage: bash tools/r03_final.sh <tag> [what...]
# what = tests bench benchall prof pmc sq alu   (default: bench prof pmc alu)
tag=$1; shift; what="${*:-bench prof pmc alu}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then ( cd $R && timeout ${TEST_TIMEOUT:-900} python -m pytest tests -q -x -m gpu --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -22 $O/pytest.log ); fi
if has bench; then ( cd $R && timeout 600 python bench.py --cpu-baseline ${CPU_BASELINE:-none} > $O/bench_chr1_250M_hifi30x.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench_chr1_250M_hifi30x.json ); fi
if has benchall; then ( cd $R; for wl in ${WL:-bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x}; do timeout 300 python bench.py --workload $wl --cpu-baseline none > $O/bench_$wl.json 2>> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", d['ms_per_step'], round(d['value']/1e6,2), d.get('value_resident'), d['stage_ms'], (d.get('boundary') or {}).get('stage_ms'))
PY
done ); fi
if has prof; then ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --cpu-baseline none --no-boundary --steps 2 --warmup 1 > $O/prof.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -30 $O/kernel_stats.csv | cut -c1-150; rm -rf $O/prof ); fi
if has pmc; then ( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python $R/bench.py --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O/pmc.$c.log 2>&1; done
  cd $R && python tools/pmc_summarize.py $O/pmc > $O/pmc_traffic.json; for c in FETCH_SIZE WRITE_SIZE; do f=$(find $O/pmc/$c -name "*counter_collection.csv" | head -1); python tools/pmc_slim.py "$f" > $O/pmc_$c.csv; done; rm -rf $O/pmc; python - <<PY
import json
d=json.load(open("$O/pmc_traffic.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"]*kv[1]["launches"])[:14]: print(f"{k[:60]:60s} launches {v['launches']:4d} MB/launch {v['hbm_bytes_per_launch']/1e6:10.1f} raw {v['hbm_bytes_per_launch_raw']/1e6:10.1f}")
PY
); fi
if has sq; then ( cd /tmp; timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python $R/bench.py --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O/sq.log 2>&1
  cd $R && python tools/kernel_counters.py $O/sq > $O/kernel_counters.txt; cat $O/kernel_counters.txt; mv $O/sq.json $O/kernel_counters.json; rm -rf $O/sq ); fi
if has alu; then ( cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/alu -- python $R/bench.py --workload bacterial5M_hifi30x --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O/alu.log 2>&1
  cd $R && python tools/sketch_alu.py $O/alu 150000000 sketch_unit_kernel > $O/sketch_alu.json; cat $O/sketch_alu.json; rm -rf $O/alu ); fi
du -sh $O
