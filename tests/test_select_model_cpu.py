"""The closed form of the second-level minimizer thinning (tests/sel_model.py: mz1_select_mz_h as window minima by two range queries + the heap fallback as a
partial sort) against the oracle's replay of the reference's state machine, on the repeat-rich scenarios - the specification a data-parallel sketch_select
kernel will be written from (the round-3 kernel runs one lane per read)."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle
import sel_model


@pytest.mark.parametrize("name,min_active", [("rr", 20), ("rr_heavy", 0), ("long_rr", 40), ("rr_big", 100), ("fz3", 30), ("nn", 0)])
def test_select_closed_form(name, min_active):
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    active = 0
    for r in range(rs.n):
        mz, x, cnt, pos, od, tot_l = o.sketch_pre(r)
        kept = sel_model.select_keep(x, cnt, pos, od, tot_l, int(rs.lengths[r]), sample_dist=o.opt.sample_dist, w=o.opt.rewin, k=o.opt.k)
        active += int(kept.size != x.size)
        assert kept.size == mz.shape[0], (name, r)
        assert (x[kept] == mz[:, 0]).all() and (pos[kept] == ((mz[:, 1] >> np.uint64(28)) & np.uint64(0x7ffffff))).all(), (name, r)
    print(f"[select model] {name}: {rs.n} reads, {active} thinned")
    assert active >= min_active, active
