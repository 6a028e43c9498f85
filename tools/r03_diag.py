import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from helpers import scenario_reads, scenario_oracle
from hifiasm_amd.api import Engine
name = sys.argv[1] if len(sys.argv) > 1 else "rr"
rs, okw = scenario_reads(name); o = scenario_oracle(name)
def run(env):
    for k in ("HAO_DBG_PACK_SEARCH",): os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(0, **okw); e.set_readset(rs); e.ha_ft_gen(); e.ha_pt_gen()
    s = e.overlap_batch_async(0, rs.n); d = e.deliver_wait(s)
    out = [tuple(np.array(x) for x in e.delivered_read(d, r)) for r in range(rs.n)]
    e.close(); return out
a = run({}); b = run({"HAO_DBG_PACK_SEARCH": "1"})
nbad = 0
for r in range(rs.n):
    ool, ofc, ofo, ocl = o.lchain(r)
    for tag, x in (("codes", a[r]), ("search", b[r])):
        ol, fc, fo, cl = x
        okk = ol.shape == ool.shape and (ol == ool).all() and cl.shape == ocl.shape and (cl == ocl).all()
        if not okk:
            nbad += 1
            if nbad < 6:
                print(tag, "read", r, "ol", ol.shape, ool.shape, "cl", cl.shape, ocl.shape)
                if cl.shape == ocl.shape:
                    bad = np.flatnonzero((cl != ocl).any(axis=1)); print(" first bad hits", bad[:8], cl[bad[:3]], ocl[bad[:3]])
                    # which chain: ol[:,10] = offset of the chain's hits in cl, ol[:,9] n? print the overlap rows around
                    st = ool[:, 10]; k = np.searchsorted(st, bad[0], side="right") - 1; print(" chain", k, ool[k], "hit index in chain", bad[0] - st[k])
print("bad", nbad, "of", 2 * rs.n)
