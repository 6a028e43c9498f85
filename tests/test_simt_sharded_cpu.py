from simt_suite import reexport

# (the two size tests - index positions beyond 2^32, configs[1] over eight ranks - need tens of GB / minutes of emulation; RCCL itself is not emulated)
reexport(globals(), "test_gpu_sharded", skip=("test_replicated_index_beyond_2_32", "test_loopback_world8_configs1", "test_rccl_single_rank"))
