from simt_suite import reexport

reexport(globals(), "test_gpu_zzz_edgrid")
