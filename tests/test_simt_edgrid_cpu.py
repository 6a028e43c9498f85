from simt_suite import reexport, FULL

# (default selection: a one-word band on a batch that does not start at read 0, a read set with N bases; HAO_SIMT_FULL=1: all six, with two- and three-word bands)
reexport(globals(), "test_gpu_zzz_edgrid", drop=lambda v: not FULL and isinstance(v, (tuple, list)) and tuple(v) in (("hifi", 375, 40), ("edge", 100, 3), ("hifi", 775, 70)))
