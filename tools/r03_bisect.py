import os, sys, subprocess
sys.path.insert(0, "."); sys.path.insert(0, "tests")
if len(sys.argv) > 1:
    from helpers import scenario_reads
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads("hifi")
    e = Engine(0, **okw); e.set_readset(rs)
    print("ft", e.ha_ft_gen(), flush=True)
    print("pt", e.ha_pt_gen(), flush=True)
    if os.environ.get("BIS_SKETCH"):
        print("hist", int(e.hist(1).sum()), flush=True)
        e.sketch_batch(0, rs.n); print("sketch ok", flush=True)
    e.overlap_batch(0, rs.n); print("batch ok", e.batch_totals(), flush=True)
    sys.exit(0)
for env in ({"HAO_DBG_SYNC": "1"}, {"BIS_SKETCH": "1"}, {"HAO_SEED_NOPERM": "1", "HAO_DBG_SYNC": "1"}, {"HAO_DBG_SK_NOFUSE": "1"}, {"HAO_DBG_PACK_SEARCH": "1"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "tools/r03_bisect.py", "x"], env=e, capture_output=True, text=True)
    print("==", env, "rc", r.returncode); print(r.stdout[-400:]); print(r.stderr[-1500:])
