"""TEST INFRASTRUCTURE: random small workloads and option mixes through the emulated device library (tests/simt) against the oracle - every read's minimizers, seed
hits, overlaps, fake cigars and chained hits, plus the coverage peaks.  `python tests/simt_fuzz.py SEED [SEED ...]` prints one line per case (OK / what differs);
tests/test_simt_fuzz_cpu.py runs a fixed handful."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def case(seed):
    rng = np.random.default_rng(seed)
    d = dict(genome_size=int(rng.integers(6_000, 40_000)), coverage=int(rng.integers(4, 26)), read_len=int(rng.integers(300, 7000)), err=float(rng.choice([0.0, 0.0005, 0.002, 0.01, 0.03])),
             seed=int(seed), repeat_rich=int(rng.choice([0, 0, 1, 2])))
    d["len_jit"] = int(rng.integers(0, max(1, d["read_len"] // 2)))
    if rng.random() < 0.3:
        d["n_rate"] = float(rng.choice([0.0002, 0.001, 0.005]))
    o = {}
    if rng.random() < 0.7:
        o["k"] = int(rng.integers(11, 64)); o["w"] = int(rng.integers(3, 100))
    if rng.random() < 0.25:
        o["hpc"] = 0
    if rng.random() < 0.3:
        o["is_ont"] = 1
    if rng.random() < 0.35:
        o["bf_shift"] = int(rng.integers(16, 27))
    if rng.random() < 0.25:
        o["bw_thres"] = float(rng.choice([0.001, 0.02, 0.2]))
    if rng.random() < 0.2:
        o["hg_size"] = int(d["genome_size"] * float(rng.choice([0.5, 1.0, 2.0])))
    if rng.random() < 0.15:
        o["max_n_chain"] = int(rng.integers(1, 12))
    return d, o


def run(seed):
    import simt_build
    from hifiasm_amd import api, synth
    import oracle_py
    path = simt_build.build_lib()
    api.lib_path = lambda: path; api._LIB = None
    d, okw = case(seed)
    rs = synth.dataset(**d)
    o = oracle_py.Oracle(rs.codes, rs.code_off, **okw)
    e = api.Engine(0, **okw); e.set_readset(rs)
    bad = []
    if e.ha_ft_gen() != o.ft_gen():
        bad.append("ft peak")
    hom, het = e.ha_pt_gen(); ohom = o.pt_gen(); st = o.stats()
    if (hom, het) != (ohom, st["het_cov"]):
        bad.append(f"pt peaks {(hom, het)} != {(ohom, st['het_cov'])}")
    e.sketch_batch(0, rs.n)
    nb = sum(1 for r in range(rs.n) if not np.array_equal(e.fetch_sketch(r), o.sketch(r)))
    if nb:
        bad.append(f"sketch {nb}")
    e.overlap_batch(0, rs.n)
    nh = nol = tot = 0
    for r in range(rs.n):
        a, b = e.fetch_seed_hits(r), o.seed_hits(r)
        nh += int(a.shape != b.shape or (a != b).any())
        ol, fc, fo, cl = e.h_ec_lchain(r); ool, ofc, ofo, ocl = o.lchain(r); tot += ool.shape[0]
        nol += int(not (ol.shape == ool.shape and (ol == ool).all() and fc.shape == ofc.shape and (fc == ofc).all() and (fo == ofo).all() and cl.shape == ocl.shape and (cl == ocl).all()))
    if nh:
        bad.append(f"seed hits {nh}")
    if nol:
        bad.append(f"overlaps {nol}")
    e.close()
    return d, okw, rs.n, tot, bad


if __name__ == "__main__":
    for s in sys.argv[1:]:
        d, okw, n, tot, bad = run(int(s))
        print(f"seed {s}: {'OK' if not bad else 'DIFF ' + '; '.join(bad)}  reads {n} overlaps {tot}  {d} {okw}", flush=True)
