"""BASELINE.json configs[1] at full size (5 Mb genome, 30x, 10 000 reads of 15 kb, 0.1 % error: 150 Mbases, ~119 M seed hits, ~0.6 M overlaps):
size-independent properties of the HIP path, plus a direct comparison with the oracle on a sample of reads."""
import zlib

import numpy as np
import pytest

import oracle_py

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    from hifiasm_amd import synth
    from hifiasm_amd.api import Engine
    g = synth.make_genome(5_000_000, seed=11)
    rs = synth.make_reads(g, 10_000, 15000, 0.001, seed=12)
    e = Engine(0)
    e.set_readset(rs)
    e.ha_ft_gen()
    e.ha_pt_gen()
    yield e, rs
    e.close()


def _read_crc(e, r):
    ol, fc, fo, cl = e.h_ec_lchain(r)
    c = zlib.crc32(np.ascontiguousarray(ol).tobytes())
    c = zlib.crc32(np.ascontiguousarray(fc).tobytes(), c)
    c = zlib.crc32(np.ascontiguousarray(fo).tobytes(), c)
    return zlib.crc32(np.ascontiguousarray(cl).tobytes(), c), ol.shape[0]


def test_whole_pass_properties(full):
    e, rs = full
    n = rs.n
    e.overlap_batch(0, n)
    t = e.batch_totals()
    assert t["overlaps"] > 500_000 and t["seed_hits"] > 100_000_000
    sample = list(range(0, n, 37))
    ref = {r: _read_crc(e, r) for r in sample}
    tot = 0
    for r in sample[:60]:
        kh = e.fetch_seed_hits(r)
        # seed hits: (target, strand) blocks in ascending order, query position then target offset ascending inside (anchor.cpp:1040-1076)
        key = (kh[:, 0].astype(np.uint64) & 0x7fffffff) << 1 | (kh[:, 0].astype(np.uint64) >> 31)
        assert (np.diff(key.astype(np.int64)) >= 0).all()
        same = np.diff(key.astype(np.int64)) == 0
        so = kh[:, 2].astype(np.int64); of = kh[:, 1].astype(np.int64)
        assert ((np.diff(so) > 0) | ((np.diff(so) == 0) & (np.diff(of) >= 0)) | ~same).all()
        ol, fc, fo, cl = e.h_ec_lchain(r)
        tot += ol.shape[0]
        if ol.shape[0]:
            L = int(rs.lengths[r])
            assert (ol[:, 0] == r).all() and (ol[:, 4] != r).all()                      # x_id is the query, never overlaps itself
            assert (ol[:, 1] <= ol[:, 2]).all() and (ol[:, 2] < L).all()
            assert (ol[:, 5] <= ol[:, 6]).all() and (ol[:, 6] < rs.lengths[ol[:, 4]]).all()
            xs = ol[:, 1].astype(np.int64) << 32 | ol[:, 2].astype(np.int64)
            assert (np.diff(xs) >= 0).all()                                              # ol->list order: (x_pos_s, x_pos_e)
            assert (ol[:, 9] == 0).all()                                                 # align_length zeroed on return (anchor.cpp:2098)
            # a chain's hits: ordinal tag = its position before the final sort; colinear: both coordinates strictly increasing
            starts = ol[:, 10].astype(np.int64)
            for s0 in starts[:8]:
                tag = cl[s0, 0] & 0x7fffffff
                m = s0
                while m < cl.shape[0] and (cl[m, 0] & 0x7fffffff) == tag:
                    m += 1
                run = cl[s0:m]
                assert (np.diff(run[:, 2].astype(np.int64)) > 0).all() and (np.diff(run[:, 1].astype(np.int64)) > 0).all()
    assert tot > 0
    # idempotence: the same pass again
    e.overlap_batch(0, n)
    assert all(_read_crc(e, r) == ref[r] for r in sample)
    # batch-split invariance: checksum of checksums over three unequal sub-batches
    cuts = [0, n // 3, n // 3 + 1777, n]
    seen = {}
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e.overlap_batch(lo, hi)
        for r in sample:
            if lo <= r < hi:
                seen[r] = _read_crc(e, r)
    assert seen == ref


def test_sampled_reads_equal_the_oracle(full):
    """the C restatement builds the tables of the full set on the host (one thread, ~1 min) and chains a sample of reads"""
    e, rs = full
    o = oracle_py.Oracle(rs.codes, rs.code_off)
    assert o.ft_gen() == e.stats()["ft_peak_hom"]
    o.pt_gen()
    st, es = o.stats(), e.stats()
    assert (st["hom_cov"], st["het_cov"], st["high_occ"], st["low_occ"]) == (es["hom_cov"], es["het_cov"], es["high_occ"], es["low_occ"])
    e.overlap_batch(0, rs.n)
    for r in range(5, rs.n, 499):
        a, b = e.fetch_seed_hits(r), o.seed_hits(r)
        assert a.shape == b.shape and (a == b).all(), r
        ol, fc, fo, cl = e.h_ec_lchain(r)
        ool, ofc, ofo, ocl = o.lchain(r)
        assert ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and (fo == ofo).all() and (cl == ocl).all(), r
