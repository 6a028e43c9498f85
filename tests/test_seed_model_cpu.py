"""The seed stage's formulation (tests/seed_model.py: generation order + stable partition by (target, strand) + the reversal rule for opposite-strand hits, anchor →
minimizer through the boundary mask of a 64-anchor window) against the oracle's restatement of minimizers_qgen0 (materialised anchors, sorted), on the CPU."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle
import seed_model


@pytest.mark.parametrize("name,step", [("hifi", 3), ("rr", 5), ("nn", 4), ("ont", 3), ("edge", 3), ("rr_heavy", 40)])
def test_bins_instead_of_digits(name, step):
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    keys, off, pos = o.pt_table()
    st = o.stats()
    wgt = seed_model.weight_table(st["high_occ"], st["low_occ"])
    n_hits = n_rev_runs = 0
    for r in range(0, rs.n, step):
        want = o.seed_hits(r)
        got = seed_model.seed_hits_model(o.sketch(r), keys, off, pos, rs.lengths, wgt, r)
        assert got.shape == want.shape, (name, r, got.shape, want.shape)
        assert (got == want).all(), (name, r, np.flatnonzero((got != want).any(axis=1))[:5])
        n_hits += want.shape[0]
        w = want[want[:, 0] >> 31 == 1]      # opposite-strand hits of one minimizer in one target: the reversal rule had something to do
        n_rev_runs += int(((w[1:, 0] == w[:-1, 0]) & (w[1:, 2] == w[:-1, 2])).sum()) if w.shape[0] > 1 else 0
    print(f"[seed model] {name}: {n_hits} hits, {n_rev_runs} opposite-strand hits that share minimizer and target with their predecessor")
    assert n_hits > 1000


def test_window_boundaries():
    """hao_seed_locate's arithmetic on its own: boundaries at the window's first and last positions, at position 64 (the next window's first anchor), 63 of them in one window"""
    rng = np.random.default_rng(3)
    for trial in range(200):
        nk = int(rng.integers(1, 300))
        cnt = rng.integers(1, 6 if trial % 3 else 200, nk)
        if trial % 7 == 0:
            cnt[:] = 1                                   # a new minimizer at every anchor
        ao = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        n = int(ao[-1]); kc = 0
        for x0 in range(0, n, 64):
            k, kc2 = seed_model.locate_window(ao, nk, kc, x0)
            xs = np.arange(x0, min(n, x0 + 64))
            want = np.searchsorted(ao, xs, side="right") - 1
            assert (k[:xs.size] == want).all(), (trial, x0)
            if x0 + 64 < n:
                assert kc2 == np.searchsorted(ao, x0 + 64, side="right") - 1, (trial, x0)
            kc = kc2
