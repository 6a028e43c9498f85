from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_tables", keep=("hifi", "nn", "bf22", "hg2", "edge", "fz0"))
