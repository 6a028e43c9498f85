# Round-4 measurement legs (on the GPU box).  usage: bash tools/r04_run.sh <tag> [legs...]
# legs: tests testsnew bench benchall benchrr prof profb pmc seedctr ed edprof alu
tag=$1; shift; what="${*:-bench}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }
WLMAIN=${WLMAIN:-chr1_250M_hifi30x}
if has testsnew; then ( cd $R && timeout ${TEST_TIMEOUT:-900} python -m pytest tests/test_gpu_zz_new.py tests/test_gpu_ed.py tests/test_gpu_altpaths.py -q -m gpu --durations=8 -rxXfs > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log; tail -40 $O/pytest_new.log ); fi
if has tests; then ( cd $R && timeout ${TEST_TIMEOUT:-1200} python -m pytest tests -q -m gpu --durations=15 -rxXfs > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -45 $O/pytest.log ); fi
if has ed; then ( cd /tmp; for w in 0 1 2; do timeout 300 python $R/tools/bench_ed.py hifi_15k 400 --wide $w; done > $O/bench_ed.txt 2>&1; cat $O/bench_ed.txt ); fi
if has edprof; then ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/edprof -- python $R/tools/bench_ed.py hifi_15k 400 --wide 0 > $O/edprof.log 2>&1
  f=$(find $O/edprof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_ed.csv; head -16 $O/kernel_stats_ed.csv | cut -c1-170; rm -rf $O/edprof ); fi
if has bench; then ( cd $R && timeout 900 python bench.py --cpu-baseline ${CPU_BASELINE:-none} ${BENCH_ARGS:-} > $O/bench_$WLMAIN.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3500 $O/bench_$WLMAIN.json ); fi
if has benchall; then ( cd $R; for wl in ${WL:-bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x}; do timeout 300 python bench.py --workload $wl --cpu-baseline none --no-variants > $O/bench_$wl.json 2>> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", d['ms_per_step'], round(d['value']/1e6,2), d.get('ms_per_step_resident'), d['stage_ms'], (d.get('boundary') or {}).get('stage_ms'))
PY
done ); fi
if has benchrr; then ( cd $R && timeout 900 python bench.py --workload chr1_250M_hifi30x_repeat --cpu-baseline none --no-variants ${BENCH_ARGS:-} > $O/bench_chr1_250M_hifi30x_repeat.json 2>> $O/bench.err; echo "benchrr rc=$?"; tail -c 3000 $O/bench_chr1_250M_hifi30x_repeat.json ); fi
if has prof; then ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 2 --warmup 1 > $O/prof.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -30 $O/kernel_stats.csv | cut -c1-150; rm -rf $O/prof ); fi
if has profb; then ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-variants --steps 2 --warmup 1 > $O/profb.log 2>&1
  f=$(find $O/profb -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_boundary.csv; head -30 $O/kernel_stats_boundary.csv | cut -c1-150; rm -rf $O/profb ); fi
if has pmc; then ( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/pmc.$c.log 2>&1; done
  cd $R && python tools/pmc_summarize.py $O/pmc > $O/pmc_traffic.json; for c in FETCH_SIZE WRITE_SIZE; do f=$(find $O/pmc/$c -name "*counter_collection.csv" | head -1); python tools/pmc_slim.py "$f" > $O/pmc_$c.csv; done; rm -rf $O/pmc; python - <<PY
import json
d=json.load(open("$O/pmc_traffic.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"]*kv[1]["launches"])[:14]: print(f"{k[:60]:60s} launches {v['launches']:4d} MB/launch {v['hbm_bytes_per_launch']/1e6:10.1f} raw {v['hbm_bytes_per_launch_raw']/1e6:10.1f}")
PY
); fi
if has seedctr; then ( cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sctr/p1 -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/sctr1.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ATOMIC_RETURN TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --output-format csv -d $O/sctr/p2 -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/sctr2.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_SCA TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/sctr/p3 -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/sctr3.log 2>&1
  cd $R && python tools/pmc_kernels.py $O/sctr seed_bin chain_group hao_index_finish sketch_unit > $O/seed_counters.txt 2>&1; mv $O/sctr.json $O/seed_counters.json; cat $O/seed_counters.txt; tail -3 $O/sctr1.log $O/sctr2.log $O/sctr3.log; rm -rf $O/sctr ); fi
if has alu; then ( cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/alu -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/alu.log 2>&1
  cd $R && python tools/sketch_alu.py $O/alu 7500002354 sketch_unit_kernel > $O/sketch_alu.json; cat $O/sketch_alu.json; rm -rf $O/alu ); fi

if has edgrid; then ( cd /tmp; for w in 0 1 2; do timeout 300 python $R/tools/bench_ed.py hifi_15k 400 --grid --wide $w; done > $O/bench_ed_grid.txt 2>&1; cat $O/bench_ed_grid.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/edprofg -- python $R/tools/bench_ed.py hifi_15k 400 --grid --wide 0 > $O/edprofg.log 2>&1
  f=$(find $O/edprofg -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_ed_grid.csv; head -6 $O/kernel_stats_ed_grid.csv | cut -c1-60,200-330; rm -rf $O/edprofg ); fi
du -sh $O
