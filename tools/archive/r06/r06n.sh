# round 6: the list-major seed kernel with its record loads staggered through the merge (HAO_SEED_LDS_STG=1) against all loads before the merge, one box
cd $GRAFT_REPO_ROOT
bash tools/r05_ab.sh r06n "tests/test_gpu_overlap.py" "" \
  "base::chr1_250M_hifi30x:--steps 5 --no-boundary" "stg:HAO_SEED_LDS_STG=1:chr1_250M_hifi30x:--steps 5 --no-boundary" \
  "base2::chr1_250M_hifi30x:--steps 5 --no-boundary" "stg2:HAO_SEED_LDS_STG=1:chr1_250M_hifi30x:--steps 5" 
