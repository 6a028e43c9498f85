"""TEST INFRASTRUCTURE: one scenario through the emulated device library (tests/simt) against the oracle, as a script - so that a test can run it in a fresh
process with another emulator setting (HAO_SIMT_ASCENDING=1: the lanes of a wave run from the lowest up instead of from the highest down).
`python tests/simt_pipeline.py SCENARIO` prints `OK <reads> <overlaps>` or what differs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(name):
    import simt_build
    from hifiasm_amd import api
    from helpers import scenario_reads, scenario_oracle
    path = simt_build.build_lib()
    api.lib_path = lambda: path; api._LIB = None
    rs, okw = scenario_reads(name); o = scenario_oracle(name)
    e = api.Engine(0, **okw); e.set_readset(rs)
    bad = []
    if e.ha_ft_gen() != o.stats()["ft_peak_hom"]:
        bad.append("ft peak")
    if e.ha_pt_gen() != (o.stats()["hom_cov"], o.stats()["het_cov"]):
        bad.append("pt peaks")
    e.overlap_batch(0, rs.n)
    tot = 0
    for r in range(rs.n):
        a, b = e.fetch_seed_hits(r), o.seed_hits(r)
        if a.shape != b.shape or (a != b).any():
            bad.append(f"seed hits of read {r}")
        ol, fc, fo, cl = e.h_ec_lchain(r); ool, ofc, ofo, ocl = o.lchain(r); tot += ool.shape[0]
        if not (ol.shape == ool.shape and (ol == ool).all() and fc.shape == ofc.shape and (fc == ofc).all() and (fo == ofo).all() and cl.shape == ocl.shape and (cl == ocl).all()):
            bad.append(f"overlaps of read {r}")
    e.close()
    print("OK" if not bad else "DIFF " + "; ".join(bad[:6]), rs.n, tot)
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
