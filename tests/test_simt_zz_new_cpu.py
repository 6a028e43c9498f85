from simt_suite import reexport, FULL

# (the thinning kernels run on the repeat-rich sets only - the full selection; the full-size fixture is a GPU test)
reexport(globals(), "test_gpu_zz_new", skip=("test_thinning_kernels_full_size",) + (() if FULL else ("test_thinning_kernels",)), keep=("hifi",))
