"""hao_batch_digest (device) == digest of what the fetch path returns == digest computed from the oracle's results, on small scenarios;
and the final-round arguments (bw_thres = 0.001, ecovlp.cpp:3957) through hao_overlap_batch_ex."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle, digest_result, digest_hits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["hifi", "rr", "edge", "bw001"])
def test_digest_matches_fetch_and_oracle(name):
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    e.overlap_batch(0, rs.n)
    d, k = e.batch_digest(rs.n)
    for r in range(rs.n):
        ol, fc, fo, cl = e.h_ec_lchain(r)
        ool, ofc, ofo, ocl = o.lchain(r)
        assert digest_result(ol, fc, cl) == d[r] == digest_result(ool, ofc, ocl), r
        assert digest_hits(e.fetch_seed_hits(r)) == k[r] == digest_hits(o.seed_hits(r)), r
    e.close()


def test_bw_override_per_pass():
    """one engine, two passes: EC-round arguments, then the final-round bw_thres on the same index"""
    import oracle_py
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads("ont")
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    o2 = oracle_py.Oracle(rs.codes, rs.code_off, **dict(okw, bw_thres=0.001))
    o2.ft_gen(); o2.pt_gen()
    e.overlap_batch(0, rs.n, bw_thres=0.001)
    n_diff = 0
    o1 = scenario_oracle("ont")
    for r in range(rs.n):
        ol, fc, fo, cl = e.h_ec_lchain(r)
        ool, ofc, ofo, ocl = o2.lchain(r)
        assert ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and (cl == ocl).all(), r
        a = o1.lchain(r)[0]
        n_diff += int(a.shape != ool.shape or (a != ool).any())
    assert n_diff > 0          # the narrower band does change results on 1 % error reads (else the test proves nothing)
    e.close()
