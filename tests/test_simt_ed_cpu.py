from simt_suite import reexport, FULL

reexport(globals(), "test_gpu_ed", keep=("hifi",), drop=lambda v: not FULL and isinstance(v, (tuple, list)) and True in [x is True for x in v])      # (True: the two-word bands)
