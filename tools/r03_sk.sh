# round 3: unit sketch kernel - parity (sketch / tables / edge / full-size samples) + timing + SQ counters of the sketch kernel
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sketch.py tests/test_gpu_edge.py tests/test_gpu_tables.py tests/test_gpu_altpaths.py tests/test_gpu_fullgold.py -x -q -m gpu --deselect tests/test_gpu_fullgold.py::test_chr1_bloom_f37 > gpurun_out/sk_t.log 2>&1
tail -5 gpurun_out/sk_t.log
for v in ${SK_VARIANTS:-new}; do
  if [ $v = old ]; then export HAO_DBG_SK_V2=1; else unset HAO_DBG_SK_V2; fi
  for wl in chr1_250M_hifi30x bacterial5M_hifi30x_repeat; do
    timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-boundary --steps 3 --warmup 1 2>gpurun_out/sk_b_${v}_$wl.err | tail -1 > gpurun_out/sk_b_${v}_$wl.json
    python -c "
import json,sys
d=json.loads(open('gpurun_out/sk_b_${v}_$wl.json').read()); print('$v', d['config']['workload'], round(d['value']/1e6,2), d['ms_per_step'], d['stage_ms'])"
  done
done
if [ -n "$SK_COUNTERS" ]; then
  R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sk_sq; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O -- python $R/bench.py --cpu-baseline none --no-boundary --steps 1 --warmup 0 > $O.log 2>&1
  cd $R; python tools/sketch_alu.py $O 7500002354 sketch_unit_kernel > gpurun_out/sketch_alu_unit.json; cat gpurun_out/sketch_alu_unit.json
  find $O -name "*.csv" -size +20M -delete
fi
