# round 6: the whole -m gpu suite at the list-major seed kernel's state, the corrected copy yardstick, and the repeat-rich twin with the list-major kernel forced on
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -14 $O/pytest_gpu.log
timeout 300 tools/ubench_gather > $O/ubench_gather.txt 2>&1; cat $O/ubench_gather.txt
bash tools/r05_ab.sh r06m "" "" \
  "rr_tables::chr1_250M_hifi30x_repeat:--steps 2 --no-boundary" "rr_lds:HAO_SEED_MERGE_AVG=100000:chr1_250M_hifi30x_repeat:--steps 2 --no-boundary" \
  "rr_lds_maxn:HAO_SEED_MERGE_AVG=100000,HAO_SEED_MERGE_MAXN=100000:chr1_250M_hifi30x_repeat:--steps 2 --no-boundary" \
  "bacrr_tables::bacterial5M_hifi30x_repeat:--steps 5" "bacrr_lds:HAO_SEED_MERGE_AVG=100000,HAO_SEED_MERGE_MAXN=100000:bacterial5M_hifi30x_repeat:--steps 5"
