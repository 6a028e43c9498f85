from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_sketch", keep=("hifi", "ont", "nn", "k40", "hpc0", "bf22", "edge", "fz2", "fz5", "low"))
