import sys, os, threading
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import scenario_reads, scenario_oracle
from test_gpu_sharded import _shard
from hifiasm_amd.api import Engine, lib
rs, okw = scenario_reads("hifi"); o = scenario_oracle("hifi")
world = 2
cuts = [rs.n * i // world for i in range(world + 1)]
grp = lib().hao_loop_create(world)
res = [None] * world
def run(rank):
    lo, hi = cuts[rank], cuts[rank + 1]
    e = Engine(0, **okw)
    e.set_readset(_shard(rs, lo, hi)); e.set_shard(lo, rs.lengths); e.dist_init_loopback(grp, rank)
    e.ha_ft_gen(); e.ha_pt_gen()
    res[rank] = e.pt_table()
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
ok_, oo, op = o.pt_table()
for r in range(world):
    k, of, p = res[r]
    print("rank", r, "keys", k.shape, ok_.shape, "eq" if k.shape == ok_.shape and (k == ok_).all() else "DIFF", "off", "eq" if of.shape == oo.shape and (of == oo).all() else "DIFF", "pos", p.shape, op.shape, "eq" if p.shape == op.shape and (p == op).all() else "DIFF")
    if k.shape == ok_.shape and not (k == ok_).all():
        i = int(np.nonzero(k != ok_)[0][0]); print(" first key diff at", i, hex(int(k[i])), hex(int(ok_[i])), "sorted?", bool((np.diff(k.astype(np.uint64)) > 0).all()))
