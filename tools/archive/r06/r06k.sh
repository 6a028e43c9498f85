# round 6: wave 0's share of a read's hits in the list-major seed kernel (it needs no search for its start), A/B on one box
cd $GRAFT_REPO_ROOT
bash tools/r05_ab.sh r06k "" "" \
  "w144:HAO_SEED_LDS_W0=144:chr1_250M_hifi30x:--steps 5 --no-boundary" "w176:HAO_SEED_LDS_W0=176:chr1_250M_hifi30x:--steps 5 --no-boundary" \
  "w208:HAO_SEED_LDS_W0=208:chr1_250M_hifi30x:--steps 5 --no-boundary" "w240:HAO_SEED_LDS_W0=240:chr1_250M_hifi30x:--steps 5 --no-boundary" \
  "old:HAO_SEED_LDS=0:chr1_250M_hifi30x:--steps 5 --no-boundary" \
  "dbg:HAO_DBG_SEEDPHASE=1:chr1_250M_hifi30x:--steps 2 --no-boundary"
grep "seed lds" gpurun_out/r06k/dbg.err | tail -2
