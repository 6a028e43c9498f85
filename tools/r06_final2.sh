# round 6, last measurement set (one code state): the legs of tools/r06_final.sh + the sketch kernel's instruction counts from the SQ counter pass
# usage (GPU box): bash tools/r06_final2.sh <tag>
tag=${1:-r06fin9}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
bash tools/r06_final.sh $tag tests bench benchall prof profrr pmc seedctr
python - $O/seed_counters.json > $O/sketch_alu.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = [x for x in d if x.startswith("sketch_unit_kernel")][0]; c = d[k]
bases = 7500002354.0      # configs[2]: hifiasm_amd/workloads.py chr1_250M_hifi30x (bench.py prints it as sketch.bases)
out = {"kernel": "sketch_unit_kernel", "instance": k, "dispatches": c.get("dispatches"), "bases": bases, "counters_sum": {x: c[x] for x in c if x.startswith("SQ_")},
       "valu_wave_insts_per_base": c["SQ_INSTS_VALU"] / bases, "salu_wave_insts_per_base": c["SQ_INSTS_SALU"] / bases,
       "sq_wait_inst_any_over_wave_cycles": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "sq_wait_any_over_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
       "source": "rocprofv3 --pmc passes of tools/r06_final.sh seedctr (one step of bench.py, no variants), summed over the pass's sketch_unit_kernel dispatches"}
print(json.dumps(out, indent=1))
PY
cat $O/sketch_alu.json | head -30
./tools/ubench_valu > $O/ubench_valu.txt 2>&1; tail -12 $O/ubench_valu.txt
