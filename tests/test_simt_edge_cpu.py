from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_edge")
