# round 6, first device run of the list-major seed kernel (hao_query5.cuh): quick parity, then A/B on one box
cd $GRAFT_REPO_ROOT
PROF_WL=chr1_250M_hifi30x bash tools/r05_ab.sh r06a "tests/test_gpu_overlap.py tests/test_gpu_altpaths.py tests/test_gpu_edge.py" "" \
  "lds:HAO_SEED_LDS=1:chr1_250M_hifi30x:--steps 5" "old:HAO_SEED_LDS=0:chr1_250M_hifi30x:--steps 5" \
  "ontlds:HAO_SEED_LDS=1:ont5M_30x:--steps 5" "ontold:HAO_SEED_LDS=0:ont5M_30x:--steps 5" \
  "baclds:HAO_SEED_LDS=1:bacterial5M_hifi30x:--steps 5" "bacold:HAO_SEED_LDS=0:bacterial5M_hifi30x:--steps 5"
