"""FULL-SIZE bench workloads against the REAL reference (fixtures: tests/golden/<workload>[_f<bits>].npz, produced by oracle/_ref/ref_harness =
unmodified hifiasm on the same reads, tests/golden/make_golden_big.py):

  * chr1_250M_hifi30x = BASELINE.json configs[2] (250 Mb, 500 000 reads of 15 kb, 7.5 Gbases, ~6 G seed hits, ~30 M overlaps), at two batch
    sizes: 32 000 reads and the 62 500 reads bench.py runs with (~7.5e8 seed hits per batch, near the 2^32 / 8e8 sizing edge);
  * chr1_250M_hifi30x at -f37 (the reference's default Bloom filter, 28-bit block ids): the all-k-mer histogram, its peak and the filter table
    of ha_ft_gen's per-block replay on 5.6 G k-mer occurrences, then the same thresholds and a slice of the pass;
  * chr1_250M_hifi30x_repeat: configs[2] with the SURVEY 8d repeat recipe (half of the genome in 25-copy families + tandem arrays): the realistic case;
  * chr1_250M_hifi30x_jitter (round 6): configs[2]'s bases in reads of 8 - 25 kb (uniform): reads that are not all alike - up to 720 minimizers and 20 000 seed
    hits per read through the list-major seed kernel, the longest ones through its hand-over to the table kernels;
  * ont50M_30x / ont5M_30x (--ont mode: 30 kb reads at 1 % error, bw 0.05: the chain DP path) and bacterial5M_hifi30x_repeat (repeat families +
    tandem arrays: filter table, minimizer thinning, max_n_chain pruning).

Compared: coverage peaks / occurrence thresholds / max_n_chain, the minimizer count histogram, the totals of the pass, a digest of EVERY read's
(ol, fake cigars, cl) and of every read's seed hits (hao_batch_digest, folded over blocks of 256 reads), and 256 sampled reads verbatim
(minimizers, ol->list, fake cigars).  A slice cut differently must give the same per-read digests (batch-split invariance).  The STREAMING pass
(the one bench.py's `value` is quoted on) is compared through the bytes that landed in host memory: test_every_read_through_the_wire_format."""
import zlib

import numpy as np
import pytest

import os

from helpers import load_golden, fold_digests, digest_result, digest_hits, GOLDEN

pytestmark = pytest.mark.gpu
#        workload, fixture suffix, engine options, batch sizes of the all-reads pass
CASES = {
    "chr1": ("chr1_250M_hifi30x", "", {}, (32_000, 62_500)),
    "chr1rr": ("chr1_250M_hifi30x_repeat", "", {}, (40_000,)),
    "chr1jit": ("chr1_250M_hifi30x_jitter", "", {}, (56_000,)),
    "ont50M": ("ont50M_30x", "", {"is_ont": 1}, (12_500,)),
    "ont5M": ("ont5M_30x", "", {"is_ont": 1}, (5_000,)),
    "repeat5M": ("bacterial5M_hifi30x_repeat", "", {}, (10_000, 3_333)),
}
# batch size of the STREAMING pass (hao_overlap_batch_async): what bench.py runs - ~8e8 seed hits per batch, a pass that fits one batch cut in two
STREAM_BATCH = {"chr1": 62_500, "chr1rr": 40_000, "chr1jit": 76_000, "ont50M": 25_000, "ont5M": 2_500, "repeat5M": 5_000}
_READS = {}


def _reads(name):
    from hifiasm_amd.workloads import workload_reads
    if name not in _READS:
        _READS[name] = workload_reads(name)
    return _READS[name]


def _open(name, suffix, opts):
    from hifiasm_amd.api import Engine
    g = load_golden(name + suffix)
    rs = _reads(name)
    assert zlib.crc32(rs.lengths.tobytes()) == int(g["len_crc"][0]) and zlib.crc32(rs.packed[: 1 << 20].tobytes()) == int(g["len_crc"][1]), \
        "the synthetic read generator drifted: regenerate the fixture"
    e = Engine(0, **opts)
    e.set_readset(rs)
    hom_ft = e.ha_ft_gen()
    hom, het = e.ha_pt_gen()
    return e, rs, g, (hom_ft, hom, het)


@pytest.fixture(scope="module", params=list(CASES))
def full(request):
    name, suffix, opts, batches = CASES[request.param]
    if not os.path.exists(os.path.join(GOLDEN, name + suffix + ".npz")):
        pytest.skip(f"no fixture tests/golden/{name}{suffix}.npz (tests/golden/make_golden_big.py)")
    e, rs, g, cov = _open(name, suffix, opts)
    e.case = request.param
    yield e, rs, g, cov, batches
    e.close()


def _check_thresholds(e, rs, g, cov):
    hom_ft, hom, het = cov
    m = g["meta"]
    assert rs.n == m["n_reads"]
    assert hom_ft == m["hom_cov_ft"] and (hom, het) == (m["hom_cov"], m["het_cov"])
    st = e.stats()
    assert (st["high_occ"], st["low_occ"], st["max_n_chain"]) == (m["high_occ"], m["low_occ"], m["max_n_chain"])
    assert (e.hist(1) == g["pt_hist"]).all()                     # minimizer count histogram of ha_pt_gen (htab.cpp:1249-1256)
    if "ft_hist" in g:                                           # all-k-mer histogram and filter table of ha_ft_gen (5.6 G occurrences on configs[2]: beyond 2^32)
        assert (e.hist(0) == g["ft_hist"]).all()
        keys, vals = e.ft_table()
        assert keys.shape == g["ft_keys"].shape and (keys == g["ft_keys"]).all() and (vals == g["ft_vals"]).all()


def test_tables_and_thresholds(full):
    e, rs, g, cov, _ = full
    _check_thresholds(e, rs, g, cov)


def _pass(e, rs, g, batch, lo0=0, hi0=None, sampled=True):
    """all reads of [lo0, hi0) in batches: per-read digests + totals; the sampled reads verbatim"""
    hi0 = rs.n if hi0 is None else hi0
    sample = g["sample"].astype(np.int64)
    dig = np.zeros(hi0 - lo0, dtype=np.uint64); dkh = np.zeros(hi0 - lo0, dtype=np.uint64)
    tot_ol = tot_cl = 0
    bad = []
    for lo in range(lo0, hi0, batch):
        hi = min(hi0, lo + batch)
        e.overlap_batch(lo, hi)
        t = e.batch_totals()
        tot_ol += t["overlaps"]; tot_cl += t["chained_hits"]
        d, k = e.batch_digest(hi - lo)
        dig[lo - lo0:hi - lo0] = d; dkh[lo - lo0:hi - lo0] = k
        for i in np.flatnonzero((sample >= lo) & (sample < hi)) if sampled else ():
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
    return dig, dkh, tot_ol, tot_cl


def test_every_read_against_the_reference(full):
    e, rs, g, _, batches = full
    n = rs.n
    ref = None
    for bi, batch in enumerate(batches):
        dig, dkh, tot_ol, tot_cl = _pass(e, rs, g, batch, sampled=bi == 0)
        assert tot_ol == g["meta"]["pass_overlaps"] and tot_cl == g["meta"]["pass_chained_hits"]
        f, fk = fold_digests(dig), fold_digests(dkh)
        assert (fk == g["dig_kh_fold"]).all(), f"batch {batch}: seed hits differ in read blocks {np.flatnonzero(fk != g['dig_kh_fold'])[:10]}"
        assert (f == g["dig_fold"]).all(), f"batch {batch}: results differ in read blocks {np.flatnonzero(f != g['dig_fold'])[:10]}"
        if ref is not None:
            assert (dig == ref[0]).all() and (dkh == ref[1]).all()
        ref = (dig, dkh)
    # batch-split invariance: a slice cut differently must give the same per-read digests
    lo = min(123_457, n // 3); hi = min(n, lo + 9_001)
    e.overlap_batch(lo, hi)
    d2, k2 = e.batch_digest(hi - lo)
    assert (d2 == ref[0][lo:hi]).all() and (k2 == ref[1][lo:hi]).all()


def test_every_read_through_the_wire_format(full):
    """The pass bench.py's `value` times: hao_overlap_batch_async -> hao_deliver_wait with the copy of batch i under the kernels of batch i + 1.  What
    LANDED IN THE PINNED ARENA is digested on the host for every read (hao_delivery_digest: ol->list, fake cigars, and cl->list decoded out of the
    position-addressed wire format by hao_unpack_hits) and compared with the reference's digests; the sampled reads are also compared verbatim."""
    e, rs, g, _, _ = full
    n, batch = rs.n, STREAM_BATCH[e.case]
    sample = g["sample"].astype(np.int64)
    dig = np.zeros(n, dtype=np.uint64)
    tot = dict(ol=0, cl=0, exc=0, bytes=0)
    bad = []

    def consume(slot, lo, hi):
        d = e.deliver_wait(slot)
        assert (d.rid_lo, d.n_reads) == (lo, hi - lo)
        dig[lo:hi] = e.delivery_digest(d)
        tot["ol"] += int(d.n_ol); tot["cl"] += int(d.n_cl); tot["exc"] += int(d.n_exc); tot["bytes"] += int(d.bytes)
        for i in np.flatnonzero((sample >= lo) & (sample < hi)):
            r = int(sample[i])
            ol, fc, fo, cl = e.delivered_read(d, r)
            a, b = int(g["ol_off"][i]), int(g["ol_off"][i + 1])
            gol = g["ol"][a:b]; gfc = g["fc"][int(g["fc_off"][a]):int(g["fc_off"][b])]
            if not (ol.shape == gol.shape and (ol == gol).all() and fc.shape == gfc.shape and (fc == gfc).all()):
                bad.append(("ol/fc", r))
            if not (digest_result(ol, fc, cl) == dig[r] == g["dig_sample"][i, 0]):
                bad.append(("digest", r))

    pending = None
    for lo in range(0, n, batch):
        hi = min(n, lo + batch)
        slot = e.overlap_batch_async(lo, hi)
        if pending:
            consume(*pending)
        pending = (slot, lo, hi)
    consume(*pending)
    assert not bad, bad[:10]
    assert tot["ol"] == g["meta"]["pass_overlaps"] and tot["cl"] == g["meta"]["pass_chained_hits"]
    f = fold_digests(dig)
    assert (f == g["dig_fold"]).all(), f"delivered results differ in read blocks {np.flatnonzero(f != g['dig_fold'])[:10]}"
    print(f"[stream] {e.case}: {n} reads, {tot['cl']} chained hits ({tot['exc']} verbatim), {tot['bytes'] / 1e9:.2f} GB through the arena")


def test_sampled_minimizers(full):
    e, rs, g, _, _ = full
    sample = g["sample"].astype(np.int64)
    for i in range(0, sample.size, 4):
        r = int(sample[i])
        e.sketch_batch(r, r + 1)
        mz = e.fetch_sketch(r)
        gm = g["mz"][int(g["mz_off"][i]):int(g["mz_off"][i + 1])]
        assert mz.shape == gm.shape and (mz == gm).all(), r


def test_chr1_bloom_f37():
    """ha_ft_gen through the blocked Bloom filter at the reference's default -f37 on all 5.6 G k-mer occurrences of configs[2] (htab.cpp:99-116,
    826-880): histogram, peak, filter table; then the thresholds of ha_pt_gen and EVERY read of the all-reads pass."""
    e, rs, g, cov = _open("chr1_250M_hifi30x", "_f37", {"bf_shift": 37})
    try:
        assert (e.hist(0) == g["ft_hist"]).all()
        keys, vals = e.ft_table()
        assert keys.shape == g["ft_keys"].shape and (keys == g["ft_keys"]).all() and (vals == g["ft_vals"]).all()
        _check_thresholds(e, rs, g, cov)
        # the WHOLE pass behind the default filter (round 5; rounds 3 - 4 compared the first 32 000 reads): every read's digests, the totals, the sampled reads verbatim
        dig, dkh, tot_ol, tot_cl = _pass(e, rs, g, 83_334)
        assert tot_ol == g["meta"]["pass_overlaps"] and tot_cl == g["meta"]["pass_chained_hits"]
        f, fk = fold_digests(dig), fold_digests(dkh)
        assert (fk == g["dig_kh_fold"]).all(), f"seed hits differ in read blocks {np.flatnonzero(fk != g['dig_kh_fold'])[:10]}"
        assert (f == g["dig_fold"]).all(), f"results differ in read blocks {np.flatnonzero(f != g['dig_fold'])[:10]}"
    finally:
        e.close()
