"""The C restatement of the windowed bit-vector edit distance (oracle/hao_oracle.c: hao_or_window_ed) against the REAL reference's
ed_band_cal_semi_64_w_absent_diag on the same (pattern, text) intervals (tests/golden/ed.npz, tests/golden/make_golden_ed.py), and of the four
alignments with traceback (hao_or_window_trace: global, forward / backward extension, semi-global) against ed_band_cal_*_w_trace + gen_trace: error
count, end points and the cigar.  "wide" = thresholds of 32 .. 63, where the reference switches to its 128-bit functions (HA_ED_INIT(128)).  CPU only."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, ed_tasks, ed_global_tasks, ed_semi_trace_tasks, ed_ext_tasks, scenario_oracle

SETS = [("hifi", False), ("ont", False), ("nn", False), ("edge", False), ("hifi", True), ("ont", True)]
NOALN = 2**31 - 1


def _nr(wide):
    return 10 if wide else 24


@pytest.mark.parametrize("name,wide", SETS)
def test_window_ed_matches_the_reference(name, wide):
    g = np.load(os.path.join(GOLDEN, "ed.npz")); W = "w" if wide else ""
    t = ed_tasks(name, n_reads=_nr(wide), wide=wide)
    assert t.shape == g[f"{name}_{W}tasks"].shape and (t == g[f"{name}_{W}tasks"]).all(), "task generator drifted: regenerate tests/golden/ed.npz"
    res = scenario_oracle(name).window_ed(t)
    assert (res == g[f"{name}_{W}res"]).all(), np.flatnonzero((res != g[f"{name}_{W}res"]).any(axis=1))[:10]
    assert (res[:, 0] != NOALN).sum() > 300 and (res[:, 0] == NOALN).sum() > 100      # both outcomes are covered


@pytest.mark.parametrize("mode,key,gen", [(0, "g", ed_global_tasks), (3, "s", ed_semi_trace_tasks), (1, "x1", ed_ext_tasks), (2, "x2", ed_ext_tasks)])
@pytest.mark.parametrize("name,wide", SETS)
def test_window_trace_matches_the_reference(name, wide, mode, key, gen):
    g = np.load(os.path.join(GOLDEN, "ed.npz")); W = "w" if wide else ""
    t = gen(name, n_reads=_nr(wide), wide=wide)
    tk = f"{name}_{W}{'x' if key.startswith('x') else key}tasks"
    assert t.shape == g[tk].shape and (t == g[tk]).all(), "task generator drifted: regenerate tests/golden/ed.npz"
    res, cig = scenario_oracle(name).window_trace(t, cap=136, mode=mode)
    want, wcig = g[f"{name}_{W}{key}res"], g[f"{name}_{W}{key}cig"]
    diff = (res != want).any(axis=1)
    assert not diff.any(), (np.flatnonzero(diff)[:10], res[diff][:3], want[diff][:3])
    off = np.concatenate(([0], np.cumsum(want[:, 5])))
    bad = [q for q in range(t.shape[0]) if not (cig[q, :want[q, 5]] == wcig[off[q]:off[q + 1]]).all()]
    assert not bad, bad[:10]
    assert (res[:, 0] != NOALN).sum() > 300
