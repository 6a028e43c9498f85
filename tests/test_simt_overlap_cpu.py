from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_overlap", keep=("hifi", "ont", "k40", "bf22"), skip=() if FULL else ("test_sub_batches_agree",))
