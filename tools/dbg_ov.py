import sys, os, faulthandler
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import scenario_reads
from hifiasm_amd.api import Engine
rs, okw = scenario_reads(sys.argv[1] if len(sys.argv) > 1 else "hifi")
e = Engine(0, **okw); e.set_readset(rs); print("set", flush=True)
e.ha_ft_gen(); print("ft", flush=True)
e.ha_pt_gen(); print("pt", flush=True)
e.overlap_batch(0, rs.n); print("ov", e.batch_totals(), flush=True)
print(e.stage_times(), flush=True)
r = e.h_ec_lchain(0); print("fetch", r[0].shape, flush=True)
