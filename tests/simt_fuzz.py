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
        o["hg_size"] = max(1000, int(d["genome_size"] * float(rng.choice([0.5, 1.0, 2.0]))) // 1000 * 1000)      # (a multiple of 1000: the reference's --hg-size only parses k / m / g suffixed sizes, CommandLines.cpp:848-863 - every case must be runnable by ref_harness, tests/ref_fuzz.py)
    if rng.random() < 0.15:
        o["max_n_chain"] = int(rng.integers(1, 12))
    if seed >= 1000 and rng.random() < 0.7:      # (seeds below 1000 keep the cases of the first sweeps)
        d["degenerate"] = int(rng.integers(1, 1 << 30))
    return d, o


def degenerate(rs, dseed, k, w):
    """mixes hand-made trouble into a read set: reads shorter than / just around k and k + w, exact copies and reverse complements, homopolymer and short tandem
    stretches (one or few HPC bases; every minimizer identical), N runs and all-N reads, a read spliced from two far-apart places"""
    from hifiasm_amd import synth
    rng = np.random.default_rng(dseed)
    reads = [rs.codes[int(rs.code_off[i]):int(rs.code_off[i + 1])].copy() for i in range(rs.n)]
    out = []
    for r in reads:
        x = rng.random()
        if x < 0.06 and len(r) > 40:
            a = int(rng.integers(0, len(r) - 20)); L = int(rng.integers(1, 400)); r = np.concatenate([r[:a], np.full(L, int(rng.integers(0, 4)), dtype=np.uint8), r[a:]])      # homopolymer
        elif x < 0.12 and len(r) > 40:
            a = int(rng.integers(0, len(r) - 20)); per = int(rng.integers(2, 8)); unit = rng.integers(0, 4, per).astype(np.uint8)
            r = np.concatenate([r[:a], np.tile(unit, int(rng.integers(2, 200))), r[a:]])                                                                              # tandem repeat
        elif x < 0.18 and len(r) > 40:
            a = int(rng.integers(0, len(r) - 10)); r = r.copy(); r[a:a + int(rng.integers(1, 60))] = 4                                                                   # N run
        elif x < 0.21 and len(r) > 200:
            r = np.concatenate([r[:len(r) // 3], r[2 * len(r) // 3:]])                                                                                                     # spliced
        out.append(r)
        y = rng.random()
        if y < 0.04:
            out.append(r.copy())                                                                                                                                           # exact copy
        elif y < 0.08:
            out.append((3 - r[::-1]).astype(np.uint8) if (r < 4).all() else r[::-1].copy())                                                                              # reverse complement
        elif y < 0.14:
            L = int(rng.choice([1, 2, k - 1, k, k + 1, k + w - 2, k + w - 1, k + w, k + w + 1, int(rng.integers(1, 3 * (k + w)))]))
            a = int(rng.integers(0, max(1, len(r) - L))); out.append(r[a:a + max(1, L)].copy())                                                                          # short reads
        elif y < 0.15:
            out.append(np.full(int(rng.integers(1, 500)), 4, dtype=np.uint8))                                                                                             # all N
    return synth.from_codes(out)


def reads_of(seed):
    """the case's read set, its generator arguments and the engine / oracle options"""
    from hifiasm_amd import synth
    d, okw = case(seed)
    dg = d.pop("degenerate", 0)
    rs = synth.dataset(**d)
    if dg:
        rs = degenerate(rs, dg, okw.get("k", 51), okw.get("w", 51)); d["degenerate"] = dg
    return rs, d, okw


def run(seed, emulated=True):
    """emulated: point hifiasm_amd.api at tests/simt's library (the caller restores it); False: whatever api loads - libhao.so on a GPU box (tests/test_gpu_fuzz.py)"""
    from hifiasm_amd import api, synth
    import oracle_py
    if emulated:
        import simt_build
        path = simt_build.build_lib()
        api.lib_path = lambda: path; api._LIB = None
    rs, d, okw = reads_of(seed)
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
