# A/B of the seed kernels on the GPU box: usage bash tools/r04_seed.sh <tag> [notest]
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
if [ "$2" != "notest" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
fi
run() { # name env... -- args
  name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py --cpu-baseline none --no-variants "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s=d['stage_ms']; b=(d.get('boundary') or {})
    print(f"{sys.argv[2]:28s} resident {d['ms_per_step_resident']:9.2f} delivered {b.get('ms_per_step',0):9.2f} seed {s.get('q_sort_bins',0):8.2f} chain {s.get('q_chain',0):7.2f} sel {s.get('q_select',0):7.2f} asm {s.get('q_assemble',0):6.2f} roof {d['roofline']['kernel']} {d['roofline']['frac']} ok {(b.get('delivered_bytes_check') or {}).get('equal_to_reference')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  grep "^\[seed\]" $O/$name.err | tail -2
}
# (the kernels behind profiles/r04/seed_ab.txt under today's switch names: default = seed_bin_kernel<.., QL>, HAO_SEED_NOQL=1 = its generic tables (round 3's kernel),
#  HAO_SEED_V2=1 = seed_bin2_kernel, the wave-private barrier-free scatter pass, HAO_SEED_PF=0 without its prefetch)
run small_ql HAO_DBG_SEEDPHASE=1 -- --workload bacterial5M_hifi30x --no-boundary --steps 2 --warmup 1
run small_noql HAO_DBG_SEEDPHASE=1 HAO_SEED_NOQL=1 -- --workload bacterial5M_hifi30x --no-boundary --steps 2 --warmup 1
run small_v2 HAO_DBG_SEEDPHASE=1 HAO_SEED_V2=1 -- --workload bacterial5M_hifi30x --no-boundary --steps 2 --warmup 1
run main_ql X=1 -- --steps 3
run main_noql HAO_SEED_NOQL=1 -- --steps 3 --no-boundary
run main_v2 HAO_SEED_V2=1 -- --steps 3 --no-boundary
run main_v2_nopf HAO_SEED_V2=1 HAO_SEED_PF=0 -- --steps 3 --no-boundary
run rr_ql X=1 -- --workload chr1_250M_hifi30x_repeat --steps 2 --no-boundary
run rr_noql HAO_SEED_NOQL=1 -- --workload chr1_250M_hifi30x_repeat --steps 2 --no-boundary
run ont_ql X=1 -- --workload ont50M_30x --steps 2 --no-boundary
