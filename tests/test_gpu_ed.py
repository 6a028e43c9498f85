"""Windowed bit-vector edit distance on the device (hao_window_ed_batch) against the oracle (pinned to the reference by
tests/test_oracle_ed.py): window / candidate pairs formed like Correct.cpp:3897 does, both strands, clipped patterns (abs_diag), N bases,
unrelated pairs (no alignment within the threshold), degenerate lengths; and the global alignment with traceback (hao_window_trace_batch): error
count, end points and cigars."""
import numpy as np
import pytest

from helpers import ed_tasks, ed_global_tasks, ed_semi_trace_tasks, ed_ext_tasks, scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge", "rr", "hifi_15k"])
def test_window_ed(name):
    from hifiasm_amd.api import Engine, HaoError
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    t = ed_tasks(name, n_reads=40, seed=5)
    got = e.window_ed_batch(t)
    want = o.window_ed(t)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    assert (want[:, 0] != 2**31 - 1).sum() > 300
    bad = t[:1].copy(); bad[0, 2] = 10**8                       # pattern interval beyond the read: rejected, never read out of bounds
    with pytest.raises(HaoError):
        e.window_ed_batch(bad)
    e.close()


@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge", "rr", "hifi_15k"])
def test_window_trace(name):
    from hifiasm_amd.api import Engine, HaoError
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    t = ed_global_tasks(name, n_reads=40, seed=7)
    got, gcig = e.window_trace_batch(t)
    want, wcig = o.window_trace(t)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    bad = [q for q in range(t.shape[0]) if not (gcig[q, :want[q, 5]] == wcig[q, :want[q, 5]]).all()]
    assert not bad, bad[:10]
    assert (want[:, 0] != 2**31 - 1).sum() > 300 and want[:, 5].max() > 20
    # a capacity below the cigar length: the entries are counted, the result is otherwise the same
    g2, c2 = e.window_trace_batch(t, cap=4)
    assert (g2 == want).all()
    badt = t[:1].copy(); badt[0, 6] = 10**8                     # text interval beyond the read: rejected, never read out of bounds
    with pytest.raises(HaoError):
        e.window_trace_batch(badt)
    e.close()


@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge", "rr", "hifi_15k"])
def test_window_semi_trace(name):
    from hifiasm_amd.api import Engine, HaoError
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    t = ed_semi_trace_tasks(name, n_reads=40, seed=9)
    got, gcig = e.window_trace_batch(t, mode=3)
    want, wcig = o.window_trace(t, mode=3)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    bad = [q for q in range(t.shape[0]) if not (gcig[q, :want[q, 5]] == wcig[q, :want[q, 5]]).all()]
    assert not bad, bad[:10]
    assert (want[:, 0] != 2**31 - 1).sum() > 300
    # the traced variant agrees with the plain one on (err, pe)
    plain = e.window_ed_batch(t)
    assert (plain[:, 0] == got[:, 0]).all() and (plain[:, 1] == got[:, 2]).all()
    badt = t[:1].copy(); badt[0, 2] = badt[0, 6] + 2 * badt[0, 8] + 5      # the band does not cover the pattern
    with pytest.raises(HaoError):
        e.window_trace_batch(badt, mode=3)
    e.close()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge", "rr", "hifi_15k"])
def test_window_extension_trace(name, mode):
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    t = ed_ext_tasks(name, n_reads=40, seed=15)
    got, gcig = e.window_trace_batch(t, mode=mode)
    want, wcig = o.window_trace(t, mode=mode)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    bad = [q for q in range(t.shape[0]) if not (gcig[q, :want[q, 5]] == wcig[q, :want[q, 5]]).all()]
    assert not bad, bad[:10]
    assert (want[:, 0] != 2**31 - 1).sum() > 200
    e.close()
