from simt_suite import reexport, FULL

# the index sort on 40 hash bits (radix passes on bits 24 .. 64, hao_sort40_mark_kernel, the fix-up of the dirty runs) on 9 M keys: the engine takes this path
# from 2^23 minimizers on, which no emulated scenario reaches.  (The two other tests of the GPU module pin behaviours of the real rocPRIM, not of this library.)
reexport(globals(), "test_gpu_rocprim", only=("test_index_sort_on_40_bits",), drop=lambda v: v == 40_000_000 and not FULL)
