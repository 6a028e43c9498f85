cd $GRAFT_REPO_ROOT; O=gpurun_out/c16; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/pytest.log; head -3 $O/pytest.log
