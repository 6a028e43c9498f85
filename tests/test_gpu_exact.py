"""Exact-overlap check after chaining on the device (exact_ec_check, ecovlp.cpp:2803-2808 via h_ec_lchain_fast_new :5103-5131) against the
oracle (itself pinned to the reference's flags in tests/golden/*.npz): every overlap of every read, reads with N, both strands, through the blocking
fetch and through the streaming delivery."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["exact", "edge", "nn", "hifi", "rr", "ont", "long200k"])
def test_exact_flags(name):
    from hifiasm_amd.api import Engine, DELIVER_OL, DELIVER_EXACT, _arr
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    e.overlap_batch(0, rs.n)
    n_exact = n_rev = 0
    for r in range(rs.n):
        ol = e.h_ec_lchain(r)[0]
        want = o.exact(o.lchain(r)[0])
        got = e.fetch_exact(r)
        assert got.shape == want.shape and (got == want).all(), r
        n_exact += int(want.sum()); n_rev += int((want.astype(bool) & (ol[:, 7] != 0)).sum()) if ol.shape[0] else 0
    if name in ("exact", "edge", "hifi", "nn"):
        assert n_exact > 50 and n_rev > 10          # both strands are really exercised
    # the same flags through the streaming path (final-round shape: ol->list + flags, no chained hits)
    d = e.deliver_wait(e.overlap_batch_async(0, rs.n, parts=DELIVER_OL | DELIVER_EXACT))
    off = _arr(d.ol_off, rs.n + 1, np.uint64)
    fl = _arr(d.exact, int(d.n_ol), np.uint8)
    for r in range(rs.n):
        assert (fl[int(off[r]):int(off[r + 1])] == o.exact(o.lchain(r)[0])).all(), r
    e.close()
