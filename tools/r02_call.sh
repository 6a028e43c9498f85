# Round-2 GPU call: tests, bench, rocprof stats, PMC passes.  usage (on the GPU box): bash tools/r02_call.sh <tag> [what...]
# what = tests bench prof pmc alu (default: all)
tag=$1; shift; what="${*:-tests bench prof pmc alu}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then ( cd $R && timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log ); fi
if has bench; then ( cd $R && timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json; tail -5 $O/bench.err ); fi
if has prof; then ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --cpu-baseline none --no-boundary --steps 2 --warmup 1 > $O/prof.log 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; head -30 $O/kernel_stats.csv | cut -c1-160; find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -size +20M -delete ); fi
if has pmc; then ( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python $R/bench.py --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O/pmc.$c.log 2>&1; done
  cd $R && python tools/pmc_summarize.py $O/pmc > $O/pmc_traffic.json; python - <<PY
import json
d=json.load(open("$O/pmc_traffic.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"]*kv[1]["launches"])[:14]: print(f"{k[:60]:60s} launches {v['launches']:4d} MB/launch {v['hbm_bytes_per_launch']/1e6:10.1f} raw {v['hbm_bytes_per_launch_raw']/1e6:10.1f}")
PY
  find $O/pmc -name "*.csv" -size +30M -delete ); fi
if has alu; then ( cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/alu -- python $R/bench.py --workload bacterial5M_hifi30x --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O/alu.log 2>&1 \
   || timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/alu -- python $R/bench.py --workload bacterial5M_hifi30x --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O/alu.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/alu2 -- python $R/bench.py --workload bacterial5M_hifi30x --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O/alu2.log 2>&1
  cd $R && python tools/sketch_alu.py $O/alu 150000000 > $O/sketch_alu.json; cat $O/sketch_alu.json; python tools/sketch_alu.py $O/alu2 150000000 > $O/sketch_alu2.json; cat $O/sketch_alu2.json
  find $O/alu $O/alu2 -name "*.csv" -size +30M -delete ); fi
du -sh $O
