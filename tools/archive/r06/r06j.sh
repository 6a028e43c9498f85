# round 6: list-major seed kernel with histogram splitters and 16-byte record loads, phase timers, A/B on one box
cd $GRAFT_REPO_ROOT
PROF_WL=chr1_250M_hifi30x bash tools/r05_ab.sh r06j "tests/test_gpu_overlap.py tests/test_gpu_altpaths.py" "" \
  "lds:HAO_SEED_LDS=1:chr1_250M_hifi30x:--steps 5" "old:HAO_SEED_LDS=0:chr1_250M_hifi30x:--steps 5" \
  "ldsdbg:HAO_SEED_LDS=1,HAO_DBG_SEEDPHASE=1:chr1_250M_hifi30x:--steps 2 --no-boundary" \
  "ontlds:HAO_SEED_LDS=1:ont5M_30x:--steps 5" "ontold:HAO_SEED_LDS=0:ont5M_30x:--steps 5"
grep "seed lds" gpurun_out/r06j/ldsdbg.err | tail -4
