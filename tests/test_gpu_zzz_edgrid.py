"""f3 on the device end to end (hao_window_ed_grid, hao_grid.cuh): the window / candidate pairs of a batch are generated on the device from the batch's final ol->list on the
reference's window grid (WINDOW = 375, Hash_Table.h:9; Correct.cpp:5645, 3897) and aligned where they lie.  The device's task list must equal the list built on the host
from the same overlaps (helpers.ed_tasks_grid_all: query read, grid window, position in ol->list) and every result the oracle's ed_band_cal_semi_64_w_absent_diag
(pinned to the reference by tests/test_oracle_ed.py; three-word bands: the upload path's result, itself pinned to the reference's *_infi_* functions).  Thresholds of one-, two- and three-word bands; a read set with N bases; a repeat-rich one."""
import time

import numpy as np
import pytest

from helpers import ed_tasks_grid_all, scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu
NOALN = 2**31 - 1


@pytest.mark.parametrize("name,window,thre", [("hifi", 375, 15), ("hifi", 375, 40), ("nn", 375, 8), ("rr", 375, 24), ("hifi", 775, 70), ("edge", 100, 3)])
def test_grid_pairs_generated_and_aligned_on_the_device(name, window, thre):
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    try:
        e.set_readset(rs); e.ha_ft_gen(); e.ha_pt_gen()
        lo, hi = (0, rs.n) if name != "hifi" else (7, rs.n - 5)      # (a batch that does not start at read 0)
        e.overlap_batch(lo, hi)
        t0 = time.time(); n = e.window_ed_grid(window, thre); dt = time.time() - t0
        ols = [e.h_ec_lchain(r)[0] for r in range(lo, hi)]
        want_t = ed_tasks_grid_all(rs.lengths, ols, lo, window, thre)
        assert n == want_t.shape[0] and (n > 200 or name == "edge")
        got_t, got_r = e.fetch_ed_grid(n)
        assert (got_t == want_t).all(), np.flatnonzero((got_t != want_t).any(axis=1))[:10]
        # (the oracle covers bands of one and two words; three- and four-word bands are pinned to the reference's *_infi_* functions through hao_window_ed_batch, tests/test_gpu_zz_new.py)
        want_r = o.window_ed(want_t) if thre <= 63 else e.window_ed_batch(want_t)
        assert (got_r == want_r).all(), np.flatnonzero((got_r != want_r).any(axis=1))[:10]
        if thre <= 63 and n < 60_000:      # and the device-resident path agrees with the upload path on the same tasks
            assert (e.window_ed_batch(want_t) == got_r).all()
        print(f"[ed grid] {name} window {window} thre {thre}: {n} pairs, {int((want_r[:, 0] != NOALN).sum())} within the threshold, call {dt * 1e3:.2f} ms (generation + alignment, nothing crosses the host)")
    finally:
        e.close()


def test_grid_results_are_gone_after_the_scratch_is_reused():
    """hao_fetch_ed_grid after another window-alignment call or a new batch has taken the scratch: an error, not somebody else's tasks"""
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads("hifi")
    e = Engine(0, **okw)
    try:
        e.set_readset(rs); e.ha_ft_gen(); e.ha_pt_gen()
        e.overlap_batch(0, rs.n)
        n = e.window_ed_grid(375, 15)
        t, r = e.fetch_ed_grid(n)
        assert n > 200 and t.shape[0] == n
        e.window_ed_batch(t[:50])                      # the upload path reuses the task / result buffers
        with pytest.raises(Exception):
            e.fetch_ed_grid(n)
        assert e.window_ed_grid(375, 15) == n
        e.overlap_batch(3, rs.n - 3)                   # a new batch: the grid belongs to the old one's overlaps
        with pytest.raises(Exception):
            e.fetch_ed_grid(n)
    finally:
        e.close()
