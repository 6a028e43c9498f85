"""The C restatement of the windowed bit-vector edit distance (oracle/hao_oracle.c: hao_or_window_ed) against the REAL reference's
ed_band_cal_semi_64_w_absent_diag on the same (pattern, text) intervals (tests/golden/ed.npz, tests/golden/make_golden_ed.py), and of the global
alignment with traceback (hao_or_window_trace) against ed_band_cal_global_64_w_trace + gen_trace: error count, end points and the cigar.  CPU only."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, ed_tasks, ed_global_tasks, ed_semi_trace_tasks, ed_ext_tasks, scenario_oracle


@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge"])
def test_window_ed_matches_the_reference(name):
    g = np.load(os.path.join(GOLDEN, "ed.npz"))
    t = ed_tasks(name)
    assert t.shape == g[name + "_tasks"].shape and (t == g[name + "_tasks"]).all(), "task generator drifted: regenerate tests/golden/ed.npz"
    res = scenario_oracle(name).window_ed(t)
    assert (res == g[name + "_res"]).all(), np.flatnonzero((res != g[name + "_res"]).any(axis=1))[:10]
    assert (res[:, 0] != 2**31 - 1).sum() > 500 and (res[:, 0] == 2**31 - 1).sum() > 100      # both outcomes are covered


@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge"])
def test_window_trace_matches_the_reference(name):
    g = np.load(os.path.join(GOLDEN, "ed.npz"))
    t = ed_global_tasks(name)
    assert t.shape == g[name + "_gtasks"].shape and (t == g[name + "_gtasks"]).all(), "task generator drifted: regenerate tests/golden/ed.npz"
    res, cig = scenario_oracle(name).window_trace(t)
    want, wcig = g[name + "_gres"], g[name + "_gcig"]
    assert (res == want).all(), np.flatnonzero((res != want).any(axis=1))[:10]
    off = np.concatenate(([0], np.cumsum(want[:, 5])))
    bad = [q for q in range(t.shape[0]) if not (cig[q, :want[q, 5]] == wcig[off[q]:off[q + 1]]).all()]
    assert not bad, bad[:10]
    assert (res[:, 0] != 2**31 - 1).sum() > 500 and (res[:, 0] == 2**31 - 1).sum() > 100 and want[:, 5].max() > 30


@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge"])
def test_window_semi_trace_matches_the_reference(name):
    g = np.load(os.path.join(GOLDEN, "ed.npz"))
    t = ed_semi_trace_tasks(name)
    assert t.shape == g[name + "_stasks"].shape and (t == g[name + "_stasks"]).all(), "task generator drifted: regenerate tests/golden/ed.npz"
    res, cig = scenario_oracle(name).window_trace(t, mode=3)
    want, wcig = g[name + "_sres"], g[name + "_scig"]
    assert (res == want).all(), np.flatnonzero((res != want).any(axis=1))[:10]
    off = np.concatenate(([0], np.cumsum(want[:, 5])))
    bad = [q for q in range(t.shape[0]) if not (cig[q, :want[q, 5]] == wcig[off[q]:off[q + 1]]).all()]
    assert not bad, bad[:10]
    assert (res[:, 0] != 2**31 - 1).sum() > 300


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name", ["hifi", "ont", "nn", "edge"])
def test_window_extension_trace_matches_the_reference(name, mode):
    g = np.load(os.path.join(GOLDEN, "ed.npz"))
    t = ed_ext_tasks(name)
    assert t.shape == g[name + "_xtasks"].shape and (t == g[name + "_xtasks"]).all(), "task generator drifted: regenerate tests/golden/ed.npz"
    res, cig = scenario_oracle(name).window_trace(t, mode=mode)
    want, wcig = g[f"{name}_x{mode}res"], g[f"{name}_x{mode}cig"]
    assert (res == want).all(), (np.flatnonzero((res != want).any(axis=1))[:10], res[(res != want).any(axis=1)][:3], want[(res != want).any(axis=1)][:3])
    off = np.concatenate(([0], np.cumsum(want[:, 5])))
    bad = [q for q in range(t.shape[0]) if not (cig[q, :want[q, 5]] == wcig[off[q]:off[q + 1]]).all()]
    assert not bad, bad[:10]
    assert (res[:, 0] != 2**31 - 1).sum() > 300
