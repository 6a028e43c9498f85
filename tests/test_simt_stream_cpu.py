from simt_suite import reexport, FULL

# (the repeat-rich scenario with a switch, "rr+...": 70 s on the emulator - with HAO_SIMT_FULL=1 and on the device only)
reexport(globals(), "test_gpu_stream", keep=("hifi", "ont"), drop=lambda v: not FULL and isinstance(v, str) and v.startswith("rr+"))
