"""GPU parity on degenerate inputs that never occur in the seeded scenarios: an empty batch, a single read, a set without any k-mer
(every read shorter than k), and a set of identical reads.  (Ragged lengths, N runs, homopolymers: scenario "edge".)"""
import numpy as np
import pytest

import oracle_py

pytestmark = pytest.mark.gpu


def _both(rs, **okw):
    from hifiasm_amd.api import Engine
    e = Engine(0, **okw)
    e.set_readset(rs)
    o = oracle_py.Oracle(rs.codes, rs.code_off, **okw)
    return e, o


def _compare_all(e, o, rs):
    assert e.ha_ft_gen() == o.ft_gen()
    hom, het = e.ha_pt_gen()
    o.pt_gen()
    st = o.stats()
    assert (hom, het) == (st["hom_cov"], st["het_cov"])
    e.overlap_batch(0, rs.n)
    tot = 0
    for r in range(rs.n):
        a, b = e.fetch_seed_hits(r), o.seed_hits(r)
        assert a.shape == b.shape and (a == b).all(), r
        ol, fc, fo, cl = e.h_ec_lchain(r)
        ool, ofc, ofo, ocl = o.lchain(r)
        assert ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and (fo == ofo).all() and cl.shape == ocl.shape and (cl == ocl).all(), r
        tot += ool.shape[0]
    assert e.batch_totals()["overlaps"] == tot
    return tot


def test_empty_batch():
    from hifiasm_amd import synth
    rs = synth.dataset(genome_size=20_000, coverage=10, read_len=3000, err=0.002, seed=4)
    e, o = _both(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    e.overlap_batch(3, 3)
    t = e.batch_totals()
    assert t["overlaps"] == 0 and t["seed_hits"] == 0
    e.overlap_batch(0, rs.n)          # the engine is still usable
    assert e.batch_totals()["overlaps"] > 0
    e.close()


def test_single_read():
    from hifiasm_amd import synth
    g = synth.make_genome(10_000, seed=3)
    rs = synth.from_codes([g[100:4100].copy()])
    e, o = _both(rs)
    assert _compare_all(e, o, rs) == 0
    e.close()


def test_no_kmers_at_all():
    from hifiasm_amd import synth
    g = synth.make_genome(10_000, seed=3)
    rs = synth.from_codes([g[i * 100:i * 100 + L].copy() for i, L in enumerate((1, 7, 30, 50, 49))])
    e, o = _both(rs)
    assert _compare_all(e, o, rs) == 0
    e.close()


def test_identical_reads():
    from hifiasm_amd import synth
    g = synth.make_genome(10_000, seed=9)
    one = g[500:5500].copy()
    rs = synth.from_codes([one.copy() for _ in range(40)] + [g[2000:7000].copy() for _ in range(12)])
    e, o = _both(rs)
    assert _compare_all(e, o, rs) > 0
    e.close()


def test_bloom_with_n_reads_and_ragged_lengths():
    """the Bloom replay must skip the sentinel slots of reads with N (scalar hashing path) - scenario 'edge' at -f23"""
    from scenarios import edge_reads
    rs = edge_reads()
    e, o = _both(rs, bf_shift=23)
    assert e.ha_ft_gen() == o.ft_gen()
    import numpy as np
    assert (np.array(e.hist(0)) == o.ft_hist()).all()
    k, v = o.ft_table()
    ek, ev = e.ft_table()
    assert (ek == k).all() and (ev == v).all()
    e.close()
