from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_stream", keep=("hifi", "ont"))
