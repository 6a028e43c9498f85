import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import scenario_reads
from hifiasm_amd.api import Engine
for name in sys.argv[1:]:
    rs, okw = scenario_reads(name)
    e = Engine(0, **okw); e.set_readset(rs); e.ha_ft_gen(); e.ha_pt_gen(); e.overlap_batch(0, rs.n)
    t = e.batch_totals()
    nol = [e.h_ec_lchain(r)[0].shape[0] for r in range(rs.n)]
    nkh = [e.fetch_seed_hits(r).shape[0] for r in range(rs.n)]
    e.sketch_batch(0, rs.n)
    nmz = [e.fetch_sketch(r).shape[0] for r in range(rs.n)]
    print(name, t, "max final overlaps/read", max(nol), "max seed hits/read", max(nkh), "max mz/read", max(nmz), e.stats()["max_n_chain"], flush=True)
    e.close()
