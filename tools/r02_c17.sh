cd $GRAFT_REPO_ROOT; O=gpurun_out/c17; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_indexfile.py -x -q -m gpu 2>&1 | tail -25 > $O/pytest.log; cat $O/pytest.log
