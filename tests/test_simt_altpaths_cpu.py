from simt_suite import reexport, FULL

# (index positions beyond 2^32: tens of GB.  The switches select other kernels / host paths; the GPU suite runs them on the repeat-rich sets - 40 s each here)
_DEFAULT = ("HAO_DBG_FORCE=seq_chain", "HAO_DBG_FORCE=dp_serial", "HAO_DBG_FORCE=seq_prune", "HAO_DBG_TEST=sk_gcap=1000", "HAO_SEED_LDS=0", "HAO_SEED_LDS_RATIO=1000000", "HAO_SEED_MERGE_MAXN=3000")
reexport(globals(), "test_gpu_altpaths", skip=("test_index_positions_beyond_2_32",), replace={"name": ["rr" if FULL else "hifi"]},
         drop=lambda v: not FULL and isinstance(v, str) and v.startswith("HAO_") and v not in _DEFAULT)
