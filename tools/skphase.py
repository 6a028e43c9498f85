import sys, os
sys.path.insert(0, os.getcwd())
from hifiasm_amd import synth
from hifiasm_amd.api import Engine
g = synth.make_genome(5_000_000, seed=11); rs = synth.make_reads(g, 10000, 15000, 0.001, seed=12, want_codes=False)
e = Engine(0); e.set_readset(rs)
for it in range(3):
    e.sketch_batch(0, rs.n, use_ft=False, sample_dist=0)
print(os.environ.get("HAO_DBG_SK_PHASE"), dict(e.stage_times())["sk_chunks"])
