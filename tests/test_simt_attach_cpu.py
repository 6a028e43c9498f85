from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_attach", replace={"name": ["rr" if FULL else "hifi"]})
