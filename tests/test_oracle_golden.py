"""Pin the C restatement (oracle/hao_oracle.c) against golden dumps of the REAL
reference (tests/golden/*.npz, produced by tests/golden/make_golden.py from
oracle/_ref/ref_harness = unmodified hifiasm sources).  CPU only."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle, load_golden, crc
from scenarios import SCENARIOS

NAMES = list(SCENARIOS)


@pytest.mark.parametrize("name", NAMES)
def test_reads_regenerate_identically(name):
    rs, _ = scenario_reads(name)
    g = load_golden(name)
    assert crc(rs.codes) == int(g["codes_crc"][0])
    assert (g["rlen"] == rs.lengths).all()


@pytest.mark.parametrize("name", NAMES)
def test_counts_filter_and_index(name):
    o = scenario_oracle(name)
    g = load_golden(name)
    M = g["meta"]
    assert (o.ft_hist() == g["ft_hist"]).all()
    k, v = o.ft_table()
    rv = g["ft_vals"].astype(np.int64)
    rv[rv == 32767] = 2**31 - 1          # map value INT16_MAX is reported as INT32_MAX by ha_ft_cnt (htab.cpp:1069)
    assert (k == g["ft_keys"]).all() and (v == rv).all()
    assert (o.pt_hist() == g["pt_hist"]).all()
    pk, po, pp = o.pt_table()
    assert (pk == g["pt_keys"]).all() and (po == g["pt_off"]).all() and (pp == g["pt_pos"]).all()
    st = o.stats()
    for key in ("hom_cov", "het_cov", "max_n_chain", "high_occ", "low_occ", "ft_peak_hom", "ft_peak_het"):
        assert st[key] == M[key], key


@pytest.mark.parametrize("name", NAMES)
def test_sketch(name):
    o = scenario_oracle(name)
    g = load_golden(name)
    mz, mz0 = g["mz"].reshape(-1, 2), g["mz0"].reshape(-1, 2)
    for r in range(o.n_reads):
        a = o.sketch(r)
        b = mz[int(g["mz_off"][r]):int(g["mz_off"][r + 1])]
        assert a.shape == b.shape and (a == b).all(), f"read {r}"
        a = o.sketch(r, use_ft=False, sample_dist=0)
        b = mz0[int(g["mz0_off"][r]):int(g["mz0_off"][r + 1])]
        assert a.shape == b.shape and (a == b).all(), f"read {r} (hf=NULL)"


@pytest.mark.parametrize("name", NAMES)
def test_seed_hits_and_chains(name):
    o = scenario_oracle(name)
    g = load_golden(name)
    ol_all, fc_all = g["ol"].reshape(-1, 12), g["fc"]
    n_head = 6
    for r in range(o.n_reads):
        kh = o.seed_hits(r)
        assert kh.shape[0] == int(g["kh_off"][r + 1] - g["kh_off"][r])
        assert crc(kh) == int(g["kh_crc"][r]), f"seed hits of read {r}"
        if r < n_head:
            assert (kh == g["kh_head"][int(g["kh_off"][r]):int(g["kh_off"][r + 1])]).all()
        ol, fc, fo, cl = o.lchain(r)
        s, e = int(g["ol_off"][r]), int(g["ol_off"][r + 1])
        assert ol.shape == ol_all[s:e].shape and (ol == ol_all[s:e]).all(), f"overlap list of read {r}"
        assert (fc == fc_all[int(g["fc_off"][s]):int(g["fc_off"][e])]).all()
        assert cl.shape[0] == int(g["cl_off"][r + 1] - g["cl_off"][r])
        assert crc(cl) == int(g["cl_crc"][r]), f"chained hits of read {r}"
        assert (o.exact(ol) == g["ex"][s:e]).all(), f"exact-overlap flags of read {r}"      # exact_ec_check on the reference's own strings (ecovlp.cpp:2803, 5124-5131)
