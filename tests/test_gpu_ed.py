"""Windowed bit-vector edit distance on the device (hao_window_ed_batch) against the oracle (pinned to the reference by
tests/test_oracle_ed.py): window / candidate pairs formed like Correct.cpp:3897 does, both strands, clipped patterns (abs_diag), N bases,
unrelated pairs (no alignment within the threshold), degenerate lengths; and the four alignments with traceback (hao_window_trace_batch: global,
forward / backward extension, semi-global): error count, end points and cigars.  "wide" = thresholds of 32 .. 63 (two-word bands)."""
import numpy as np
import pytest

from helpers import ed_tasks, ed_global_tasks, ed_semi_trace_tasks, ed_ext_tasks, scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu
SETS = [("hifi", False), ("ont", False), ("nn", False), ("edge", False), ("rr", False), ("hifi_15k", False), ("hifi", True), ("ont", True), ("hifi_15k", True)]
NOALN = 2**31 - 1


@pytest.mark.parametrize("name,wide", SETS)
def test_window_ed(name, wide):
    from hifiasm_amd.api import Engine, HaoError
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    t = ed_tasks(name, n_reads=40, seed=5, wide=wide)
    got = e.window_ed_batch(t)
    want = o.window_ed(t)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    assert (want[:, 0] != NOALN).sum() > 300
    bad = t[:1].copy(); bad[0, 2] = 10**8                       # pattern interval beyond the read: rejected, never read out of bounds
    with pytest.raises(HaoError):
        e.window_ed_batch(bad)
    bad = t[:1].copy(); bad[0, 8] = 128                         # a band of 257 diagonals: not built (thre 64 .. 127: tests/test_gpu_zz_new.py)
    with pytest.raises(HaoError):
        e.window_ed_batch(bad)
    e.close()


@pytest.mark.parametrize("mode,gen", [(0, ed_global_tasks), (1, ed_ext_tasks), (2, ed_ext_tasks), (3, ed_semi_trace_tasks)])
@pytest.mark.parametrize("name,wide", SETS)
def test_window_trace(name, wide, mode, gen):
    from hifiasm_amd.api import Engine, HaoError
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    t = gen(name, n_reads=40, seed=7 + mode, wide=wide)
    if not wide and mode == 0:                                   # both band widths in one call
        t = np.concatenate([t, gen(name, n_reads=6, seed=3, wide=True)])
    got, gcig = e.window_trace_batch(t, cap=136, mode=mode)
    want, wcig = o.window_trace(t, cap=136, mode=mode)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    bad = [q for q in range(t.shape[0]) if not (gcig[q, :want[q, 5]] == wcig[q, :want[q, 5]]).all()]
    assert not bad, bad[:10]
    assert (want[:, 0] != NOALN).sum() > 200
    if mode == 0:                                               # a capacity below the cigar length: the entries are counted, the result is otherwise the same
        g2, c2 = e.window_trace_batch(t, cap=4, mode=mode)
        assert (g2 == want).all()
        badt = t[:1].copy(); badt[0, 6] = 10**8                 # text interval beyond the read: rejected, never read out of bounds
        with pytest.raises(HaoError):
            e.window_trace_batch(badt, mode=mode)
    if mode == 3:                                               # the traced variant agrees with the plain one on (err, pe); a band that does not cover the pattern is refused
        plain = e.window_ed_batch(t)
        assert (plain[:, 0] == got[:, 0]).all() and (plain[:, 1] == got[:, 2]).all()
        badt = t[:1].copy(); badt[0, 2] = badt[0, 6] + 2 * badt[0, 8] + 5
        with pytest.raises(HaoError):
            e.window_trace_batch(badt, mode=3)
    e.close()
