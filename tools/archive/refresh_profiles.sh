# regenerate the judged artefacts of a round on the GPU box: kernel stats, PMC traffic, bench lines
tag=$1
bash tools/prof.sh $tag > gpurun_out/prof_${tag}_top.txt
bash tools/pmc.sh $tag > gpurun_out/pmc_${tag}_top.txt
for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat ont5M_30x chr2M_hifi30x; do
  timeout 300 python bench.py --workload $wl 2>/dev/null | tail -1 > gpurun_out/bench_${tag}_$wl.json
done
timeout 600 python bench.py --workload chr1_250M_hifi30x --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${tag}_chr1_250M_hifi30x.json
cat gpurun_out/prof_${tag}_top.txt | head -12; cat gpurun_out/pmc_${tag}_top.txt; for f in gpurun_out/bench_${tag}_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline'], (d.get('cpu_baseline') or {}).get('value'))"; done
