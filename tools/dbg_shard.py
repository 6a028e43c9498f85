import sys, os, threading
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import scenario_reads
from test_gpu_sharded import _shard
from hifiasm_amd.api import Engine, lib
rs, okw = scenario_reads("hifi")
world = 2
cuts = [rs.n * i // world for i in range(world + 1)]
grp = lib().hao_loop_create(world)
def run(rank):
    lo, hi = cuts[rank], cuts[rank + 1]
    e = Engine(0, **okw); print("engine", rank, flush=True)
    e.set_readset(_shard(rs, lo, hi)); e.set_shard(lo, rs.lengths); e.dist_init_loopback(grp, rank); print("init", rank, flush=True)
    e.ha_ft_gen(); print("ft", rank, flush=True)
    e.ha_pt_gen(); print("pt", rank, flush=True)
    e.overlap_batch(0, hi - lo); print("ov", rank, e.batch_totals(), flush=True)
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
