# round 6: ha_ft_gen's host wall clock stage by stage (configs[2]), the drop-in executable at -f0 / -f37 on configs[1], the seed-stage switches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06o; mkdir -p $O
timeout 600 python tools/ft_time.py chr1_250M_hifi30x 2 > $O/ft_time.txt 2>&1; cat $O/ft_time.txt | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_altpaths.py tests/test_gpu_zzz_edgrid.py -q -m gpu -x --durations=5 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
