"""GPU parity: h_ec_lchain through the C ABI vs the oracle - seed hits before chaining, overlap list
(order included), fake cigars and chained hits, bit-exact for every read of every scenario."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle
from scenarios import SCENARIOS, BIG_SCENARIOS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=list(SCENARIOS) + list(BIG_SCENARIOS))
def pair(request):
    from hifiasm_amd.api import Engine
    name = request.param
    rs, okw = scenario_reads(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen()
    e.ha_pt_gen()
    e.overlap_batch(0, rs.n)
    yield name, e, scenario_oracle(name), rs
    e.close()


def test_seed_hits(pair):
    name, e, o, rs = pair
    bad = 0
    for r in range(rs.n):
        a, b = e.fetch_seed_hits(r), o.seed_hits(r)
        if a.shape != b.shape or (a != b).any():
            bad += 1
    assert bad == 0, f"{bad}/{rs.n} reads differ"


def test_overlaps(pair):
    name, e, o, rs = pair
    bad, tot = [], 0
    for r in range(rs.n):
        ol, fc, fo, cl = e.h_ec_lchain(r)
        ool, ofc, ofo, ocl = o.lchain(r)
        tot += ool.shape[0]
        ok = ol.shape == ool.shape and (ol == ool).all() and fc.shape == ofc.shape and (fc == ofc).all() and (fo == ofo).all() \
            and cl.shape == ocl.shape and (cl == ocl).all()
        if not ok:
            bad.append(r)
    assert not bad, f"{len(bad)}/{rs.n} reads differ, first {bad[:5]}"
    t = e.batch_totals()
    assert t["overlaps"] == tot


def test_sub_batches_agree(pair):
    """two half batches give the same per-read results as one batch"""
    name, e, o, rs = pair
    h = rs.n // 2
    for lo, hi in ((0, h), (h, rs.n)):
        e.overlap_batch(lo, hi)
        for r in (lo, (lo + hi) // 2, hi - 1):
            ol, fc, fo, cl = e.h_ec_lchain(r)
            ool, ofc, ofo, ocl = o.lchain(r)
            assert ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and (cl == ocl).all()
    e.overlap_batch(0, rs.n)
