"""GPU parity: minimizer sketch (K1+K3) through the C ABI vs the oracle, read by read, bit-exact."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle
from scenarios import SCENARIOS

pytestmark = pytest.mark.gpu


def _engine(name):
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    return e, rs


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_sketch_no_filter(name):
    """hf = NULL, sample_dist = 0: pure (hash) window minimizers incl. ties, first-window quirk, short reads."""
    e, rs = _engine(name)
    o = scenario_oracle(name)
    e.sketch_batch(0, rs.n, use_ft=False, sample_dist=0)
    bad = 0
    for r in range(rs.n):
        a = e.fetch_sketch(r)
        b = o.sketch(r, use_ft=False, sample_dist=0)
        if a.shape != b.shape or (a != b).any():
            bad += 1
    e.close()
    assert bad == 0, f"{bad}/{rs.n} reads differ"


def test_sketch_subrange_and_empty():
    e, rs = _engine("hifi")
    o = scenario_oracle("hifi")
    e.sketch_batch(7, 7, use_ft=False, sample_dist=0)       # empty range is legal
    e.sketch_batch(5, 23, use_ft=False, sample_dist=0)
    for r in range(5, 23):
        a, b = e.fetch_sketch(r), o.sketch(r, use_ft=False, sample_dist=0)
        assert a.shape == b.shape and (a == b).all()
    e.close()
