# round 5: the measurement legs behind profiles/r05/ on the GPU box (one code state).  usage: bash tools/r05_final.sh <tag> [legs: tests bench benchall rrab prof profrr pmc seedctr]
tag=$1; shift; what="${*:-tests bench benchall}"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; export TMPDIR=/tmp; cd $R
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }
WLMAIN=${WLMAIN:-chr1_250M_hifi30x}
if has tests; then t0=$(date +%s); timeout ${TEST_TIMEOUT:-1100} python -m pytest tests -q -m gpu --durations=12 -rxXfs > $O/pytest_gpu_last.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" >> $O/pytest_gpu_last.log; tail -22 $O/pytest_gpu_last.log; grep -h '^\[rank share\]\|^\[dropin configs1\]' $O/pytest_gpu_last.log | cut -c1-1200; fi
if has bench; then timeout 900 python bench.py --cpu-baseline sample --steps 20 --warmup 5 > $O/bench_$WLMAIN.json 2> $O/bench.err; echo "bench rc=$?"; python - $O/bench_$WLMAIN.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); v=(d.get('variants') or {}).get('repeat_rich') or {}
print('value', d['value'], 'ms', d['ms_per_step'], 'resident', d['value_resident'], d['ms_per_step_resident'], 'roofline', {k:d['roofline'][k] for k in ('kernel','achieved','frac','kernel_ms')}, [ (k['kernel'],k['kernel_ms'],k['frac']) for k in d['roofline'].get('kernels',[])])
print('stage', d['stage_ms']); print('boundary', (d.get('boundary') or {}).get('stage_ms'), (d.get('boundary') or {}).get('delivered_bytes_check'))
print('rr', v.get('value'), v.get('ms_per_step'), v.get('value_resident'), v.get('ms_per_step_resident'), (v.get('boundary') or {}).get('delivered_bytes_check')); print('cpu', d.get('cpu_baseline'))
PY
fi
if has benchall; then for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x; do timeout 300 python bench.py --workload $wl --cpu-baseline none --no-variants > $O/bench_$wl.json 2>> $O/bench.err; python - $O/bench_$wl.json $wl <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'ms', d['ms_per_step'], 'M ov/s', round(d['value']/1e6,2), 'resident', d.get('ms_per_step_resident'), round(d['value_resident']/1e6,2), 'ok', ((d.get('boundary') or {}).get('delivered_bytes_check') or {}).get('equal_to_reference'))
PY
done; fi
if has rrab; then for spec in rr_default: rr_maxn16k:HAO_SEED_MERGE_MAXN=16000 rr_all_merge:HAO_SEED_MERGE_MAXN=100000000 rr_tables:HAO_SEED_MERGE=0; do IFS=: read name envs <<< "$spec"; env ${envs:-X_=1} timeout 600 python bench.py --workload chr1_250M_hifi30x_repeat --cpu-baseline none --no-variants --no-boundary --steps 3 --warmup 1 > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']
print(sys.argv[2], 'resident', d['ms_per_step_resident'], 'seed', round(s['q_sort_bins'],1), 'chain', round(s['q_chain'],1), 'sel', round(s['q_select'],1), 'sketch', round(s['sk_chunks'],1))
PY
done; fi
if has c2ab; then for spec in c2_merge: c2_tables:HAO_SEED_MERGE=0 c2_merge_locus:HAO_SEED_LOCUS=1; do IFS=: read name envs <<< "$spec"; env ${envs:-X_=1} timeout 600 python bench.py --workload chr1_250M_hifi30x --cpu-baseline none --no-variants --no-boundary --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['stage_ms']
print(sys.argv[2], 'resident', d['ms_per_step_resident'], 'seed', round(s['q_sort_bins'],2), 'chain', round(s['q_chain'],2), 'sketch', round(s['sk_chunks'],2), 'frac', d['roofline']['frac'])
PY
done; fi
prof() { wl=$1; out=$2; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$out -- python $R/bench.py --workload $wl --cpu-baseline none --no-boundary --no-variants --steps 2 --warmup 1 > $O/prof_$out.log 2>&1; f=$(find $O/prof_$out -name "*kernel_stats.csv" | head -1); cp "$f" $O/$out.csv; head -12 $O/$out.csv | cut -c1-150; rm -rf $O/prof_$out ); }
has prof && prof $WLMAIN kernel_stats
has profrr && prof chr1_250M_hifi30x_repeat kernel_stats_repeat_rich
if has pmc; then ( cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/pmc.$c.log 2>&1; done
  cd $R && python tools/pmc_summarize.py $O/pmc > $O/pmc_traffic.json; for c in FETCH_SIZE WRITE_SIZE; do f=$(find $O/pmc/$c -name "*counter_collection.csv" | head -1); python tools/pmc_slim.py "$f" > $O/pmc_$c.csv; done; rm -rf $O/pmc; python - $O/pmc_traffic.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"]*kv[1]["launches"])[:10]: print(f"{k[:60]:60s} launches {v['launches']:4d} MB/launch {v['hbm_bytes_per_launch']/1e6:10.1f} raw {v['hbm_bytes_per_launch_raw']/1e6:10.1f}")
PY
); fi
if has seedctr; then ( cd /tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/sctr/p1 -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/sctr1.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_WAVES TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum --output-format csv -d $O/sctr/p2 -- python $R/bench.py --workload $WLMAIN --cpu-baseline none --no-boundary --no-variants --steps 1 --warmup 0 > $O/sctr2.log 2>&1
  cd $R && python tools/pmc_kernels.py $O/sctr seed_merge seed_bin chain_group sketch_unit > $O/seed_counters.txt 2>&1; mv $O/sctr.json $O/seed_counters.json; cat $O/seed_counters.txt | cut -c1-120; tail -2 $O/sctr1.log $O/sctr2.log | cut -c1-200; rm -rf $O/sctr ); fi
du -sh $O
