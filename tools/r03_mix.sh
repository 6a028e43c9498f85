# f37 at scale (replay by runs), loader / shim tests with short timeouts, profile of the repeat-rich boundary step
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gpu_fullgold.py::test_chr1_bloom_f37 -x -q -m gpu --durations=3 > gpurun_out/mix_f37.log 2>&1; tail -6 gpurun_out/mix_f37.log
timeout 300 python -m pytest tests/test_gpu_indexfile.py tests/test_gpu_tables.py tests/test_gpu_attach.py tests/test_gpu_stream.py -x -q -m gpu --durations=3 > gpurun_out/mix_a.log 2>&1; tail -6 gpurun_out/mix_a.log
HAO_SHIM_STATS=1 timeout 420 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu --durations=3 -s > gpurun_out/mix_b.log 2>&1; tail -12 gpurun_out/mix_b.log | cut -c1-300
bash tools/r03_prof.sh bacterial5M_hifi30x_repeat rr > gpurun_out/mix_prof.log 2>&1; tail -40 gpurun_out/mix_prof.log
