from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_digest", keep=("hifi",), skip=() if FULL else ("test_bw_override_per_pass",))
