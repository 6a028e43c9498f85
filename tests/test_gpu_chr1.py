"""BASELINE.json configs[2] at FULL size (250 Mb genome, 30x, 500 000 reads of 15 kb, 7.5 Gbases, ~6 G seed hits, ~30 M overlaps) against the
REAL reference: tests/golden/chr1_250M_hifi30x.npz was produced by oracle/_ref/ref_harness (unmodified hifiasm) on the same FASTA
(tests/golden/make_golden_big.py).  Compared: coverage peaks / occurrence thresholds / max_n_chain, the minimizer count histogram, the
totals of the pass, a digest of EVERY read's (ol, fake cigars, cl) and of every read's seed hits (hao_batch_digest, folded over blocks of
256 reads), and 256 sampled reads verbatim (minimizers, ol->list, fake cigars).  The pass runs in ~16 batches, so this also checks that
results do not depend on how the reads are split into batches (a second, different split of a slice must give the same digests)."""
import zlib

import numpy as np
import pytest

from helpers import load_golden, fold_digests, digest_result, digest_hits

pytestmark = pytest.mark.gpu
NAME = "chr1_250M_hifi30x"
BATCH = 32_000


@pytest.fixture(scope="module")
def chr1():
    from hifiasm_amd.workloads import workload_reads
    from hifiasm_amd.api import Engine
    g = load_golden(NAME)
    rs = workload_reads(NAME)
    assert zlib.crc32(rs.lengths.tobytes()) == int(g["len_crc"][0]) and zlib.crc32(rs.packed[: 1 << 20].tobytes()) == int(g["len_crc"][1]), \
        "the synthetic read generator drifted: regenerate the fixture"
    e = Engine(0)
    e.set_readset(rs)
    hom_ft = e.ha_ft_gen()
    hom, het = e.ha_pt_gen()
    yield e, rs, g, (hom_ft, hom, het)
    e.close()


def test_tables_and_thresholds(chr1):
    e, rs, g, (hom_ft, hom, het) = chr1
    m = g["meta"]
    assert rs.n == m["n_reads"]
    assert hom_ft == m["hom_cov_ft"] and (hom, het) == (m["hom_cov"], m["het_cov"])
    st = e.stats()
    assert (st["high_occ"], st["low_occ"], st["max_n_chain"]) == (m["high_occ"], m["low_occ"], m["max_n_chain"])
    assert (e.hist(1) == g["pt_hist"]).all()                     # minimizer count histogram of ha_pt_gen (htab.cpp:1249-1256)


def test_every_read_against_the_reference(chr1):
    e, rs, g, _ = chr1
    n = rs.n
    sample = g["sample"].astype(np.int64)
    dig = np.zeros(n, dtype=np.uint64); dkh = np.zeros(n, dtype=np.uint64)
    tot_ol = tot_cl = 0
    bad = []
    for lo in range(0, n, BATCH):
        hi = min(n, lo + BATCH)
        e.overlap_batch(lo, hi)
        t = e.batch_totals()
        tot_ol += t["overlaps"]; tot_cl += t["chained_hits"]
        d, k = e.batch_digest(hi - lo)
        dig[lo:hi] = d; dkh[lo:hi] = k
        for i in np.flatnonzero((sample >= lo) & (sample < hi)):
            r = int(sample[i])
            ol, fc, fo, cl = e.h_ec_lchain(r)
            a, b = int(g["ol_off"][i]), int(g["ol_off"][i + 1])
            gol = g["ol"][a:b]; gfc = g["fc"][int(g["fc_off"][a]):int(g["fc_off"][b])]
            if not (ol.shape == gol.shape and (ol == gol).all() and fc.shape == gfc.shape and (fc == gfc).all()):
                bad.append(("ol/fc", r))
            # the device digest, the digest of what the fetch path returns, and the reference's digest agree
            if not (digest_result(ol, fc, cl) == d[r - lo] == g["dig_sample"][i, 0]):
                bad.append(("digest", r))
            if not (digest_hits(e.fetch_seed_hits(r)) == k[r - lo] == g["dig_sample"][i, 1]):
                bad.append(("seed digest", r))
    assert not bad, bad[:10]
    assert tot_ol == g["meta"]["pass_overlaps"] and tot_cl == g["meta"]["pass_chained_hits"]
    f, fk = fold_digests(dig), fold_digests(dkh)
    assert (fk == g["dig_kh_fold"]).all(), f"seed hits differ in read blocks {np.flatnonzero(fk != g['dig_kh_fold'])[:10]}"
    assert (f == g["dig_fold"]).all(), f"results differ in read blocks {np.flatnonzero(f != g['dig_fold'])[:10]}"
    # batch-split invariance: a slice cut differently must give the same per-read digests
    lo, hi = 123_457, 123_457 + 9_001
    e.overlap_batch(lo, hi)
    d2, k2 = e.batch_digest(hi - lo)
    assert (d2 == dig[lo:hi]).all() and (k2 == dkh[lo:hi]).all()


def test_sampled_minimizers(chr1):
    e, rs, g, _ = chr1
    sample = g["sample"].astype(np.int64)
    for i in range(0, sample.size, 4):
        r = int(sample[i])
        e.sketch_batch(r, r + 1)
        mz = e.fetch_sketch(r)
        gm = g["mz"][int(g["mz_off"][i]):int(g["mz_off"][i + 1])]
        assert mz.shape == gm.shape and (mz == gm).all(), r
