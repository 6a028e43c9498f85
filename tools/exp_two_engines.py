# Experiment (not part of the product): all-reads pass with NE batch contexts (hao_attach: own stream and scratch over one index), one host thread each,
# against one context running the batches back to back.  usage: NE=2 exp_two_engines.py [workload]
import sys, os, time, threading
sys.path.insert(0, os.getcwd())
from hifiasm_amd.api import Engine
from hifiasm_amd.workloads import WORKLOADS
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "chr1_250M_hifi30x"
rs, is_ont = bench.make_reads(wl, rank=0, world=1)
NE = int(os.environ.get('NE', '2'))
e0 = Engine(0, is_ont=is_ont); e0.set_readset(rs); e0.ha_ft_gen(); e0.ha_pt_gen()
engs = [e0] + [e0.attach() for _ in range(NE - 1)]
n = rs.n
per = min(n, max(1, int(4e8 // max(1.0, 0.83 * rs.total_bases / max(1, n) * WORKLOADS[wl][1] / 30.0))))
if os.environ.get('PER'): per = int(os.environ['PER'])
ranges = [(lo, min(n, lo + per)) for lo in range(0, n, per)]
print("batches", len(ranges), "reads per batch", per, flush=True)

def run(e, rr, out):
    ov = 0
    for lo, hi in rr:
        e.overlap_batch(lo, hi); ov += e.batch_totals()["overlaps"]
    out.append(ov)

for rep in range(3):
    o1 = []; t0 = time.time(); run(engs[0], ranges, o1); t1 = time.time()
    o2 = []; th = [threading.Thread(target=run, args=(engs[k], ranges[k::NE], o2)) for k in range(NE)]
    t2 = time.time(); [t.start() for t in th]; [t.join() for t in th]; t3 = time.time()
    print(f"one context: {1e3 * (t1 - t0):.1f} ms ({sum(o1)} overlaps)   {NE} contexts, alternate batches: {1e3 * (t3 - t2):.1f} ms ({sum(o2)} overlaps)", flush=True)
    e0.ha_pt_gen()      # the views follow the rebuilt index
