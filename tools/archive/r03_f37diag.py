"""ha_ft_gen at -f37 on configs[2] with the replay's self-checks (HAO_DBG_BLOOM): stage times, histogram against the reference's"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import load_golden
from hifiasm_amd.workloads import workload_reads
from hifiasm_amd.api import Engine
os.environ["HAO_DBG_BLOOM"] = "1"
name = sys.argv[1] if len(sys.argv) > 1 else "chr1_250M_hifi30x"
t0 = time.time(); rs = workload_reads(name); print("reads", rs.n, round(time.time() - t0, 1), "s", flush=True)
g = load_golden(name + "_f37")
e = Engine(0, bf_shift=37); e.set_readset(rs)
t0 = time.time(); hom = e.ha_ft_gen(); print("ha_ft_gen", round(time.time() - t0, 2), "s  hom", hom, "golden", g["meta"]["hom_cov_ft"], flush=True)
print(e.stage_times())
h = e.hist(0); gh = g["ft_hist"]
print("distinct", int(h.sum()), int(gh.sum()), " occurrences", int((h * np.arange(4096)).sum()), int((gh * np.arange(4096)).sum()))
bad = np.flatnonzero(h != gh); print("bins that differ:", bad.size, bad[:20], h[bad[:20]], gh[bad[:20]])
e.close()
