# tests without the full-size f37 case, seed-order A/B, then the f37 diagnosis
mkdir -p gpurun_out/r03b; O=gpurun_out/r03b
timeout 600 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullgold.py::test_chr1_bloom_f37 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for v in perm noperm; do
  if [ $v = noperm ]; then export HAO_SEED_NOPERM=1; else unset HAO_SEED_NOPERM; fi
  timeout 300 python bench.py --cpu-baseline none --steps 3 --warmup 1 2>$O/bench_$v.err | tail -1 > $O/bench_$v.json
  python -c "
import json
d=json.loads(open('$O/bench_$v.json').read()); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_resident'], d['stage_ms'], d['boundary']['stage_ms'])"
done
unset HAO_SEED_NOPERM
timeout 460 python tools/r03_f37diag.py > $O/f37diag.log 2>&1; tail -40 $O/f37diag.log
