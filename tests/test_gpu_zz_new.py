"""Device code whose FIRST execution on a GPU is the driver's `pytest -m gpu` run (the file sorts last so that `pytest -x` cannot hide the rest of
the suite behind it):

  * the sketch's thinning kernel (sketch_select2_kernel): minimizers of every read and every read's h_ec_lchain result on the repeat-rich
    scenarios, plus the repeat-rich 5 Mb full-size fixture;
  * window alignment in bands of three and four 64-bit words (thre 64 .. 127; hao_al_kernel<hao_wide<3|4>, ...>) against the REFERENCE's own
    ed_band_cal_*_infi_* functions (tests/golden/ed_wide.npz, tests/golden/make_golden_ed_wide.py).

Tests of code that has never run on a device are marked xfail(strict=False) until it has (XPASS = it works)."""
import os

import numpy as np
import pytest

from helpers import (scenario_reads, scenario_oracle, load_golden, fold_digests, ed_tasks, ed_global_tasks, ed_semi_trace_tasks, ed_ext_tasks)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOALN = 2**31 - 1
SELECT_SWITCHES = [None]      # (the one-lane replay kernel of round 3 was removed in round 6: the wave kernel, hao_select2.cuh, is the only one)


def _same(a, b):
    return all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))


@pytest.mark.parametrize("switch", SELECT_SWITCHES)
@pytest.mark.parametrize("name", ["rr", "rr_big", "rr_heavy", "bf24", "fz3", "long_rr"])
def test_thinning_kernels(name, switch):
    """mz1_select_mz_h (sketch.cpp:247-330) by the wave kernel (closed form, hao_select2.cuh): the oracle's minimizers, the oracle's overlaps"""
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    if switch:
        os.environ[switch] = "1"
    try:
        e = Engine(0, **okw)
        e.set_readset(rs)
        e.ha_ft_gen(); e.ha_pt_gen()
        e.sketch_batch(0, rs.n)
        bad_mz = [r for r in range(rs.n) if not _same((e.fetch_sketch(r),), (o.sketch(r),))]
        e.overlap_batch(0, rs.n)
        bad = [r for r in range(rs.n) if not _same(e.h_ec_lchain(r), o.lchain(r))]
        e.close()
    finally:
        if switch:
            del os.environ[switch]
    assert not bad_mz, f"minimizers of {len(bad_mz)}/{rs.n} reads differ: {bad_mz[:8]}"
    assert not bad, f"{len(bad)}/{rs.n} reads differ: {bad[:8]}"


@pytest.mark.parametrize("switch", SELECT_SWITCHES)
def test_thinning_kernels_full_size(switch):
    """bacterial5M_hifi30x_repeat (10 000 reads, filter table + thinning + max_n_chain pruning) against the reference's digests of every read"""
    from hifiasm_amd.api import Engine
    from hifiasm_amd.workloads import workload_reads
    name = "bacterial5M_hifi30x_repeat"
    g = load_golden(name)
    rs = workload_reads(name)
    if switch:
        os.environ[switch] = "1"
    try:
        e = Engine(0)
        e.set_readset(rs)
        assert e.ha_ft_gen() == g["meta"]["hom_cov_ft"]
        assert e.ha_pt_gen() == (g["meta"]["hom_cov"], g["meta"]["het_cov"])
        assert (e.hist(1) == g["pt_hist"]).all()
        e.overlap_batch(0, rs.n)
        d, k = e.batch_digest(rs.n)
        e.close()
    finally:
        if switch:
            del os.environ[switch]
    assert (fold_digests(k) == g["dig_kh_fold"]).all() and (fold_digests(d) == g["dig_fold"]).all()


# ---- f3: bands of three and four words ----
def _wide_fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ed_wide.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["hifi", "ont", "nn"])
def test_wide_bands_distance(name):
    from hifiasm_amd.api import Engine, HaoError
    g = _wide_fixture()
    rs, okw = scenario_reads(name)
    t = ed_tasks(name, n_reads=12, wide=2)
    assert t.shape == g[name + "_tasks"].shape and (t == g[name + "_tasks"]).all(), "the task generator drifted: regenerate the fixture"
    assert set(np.unique((2 * t[:, 8].astype(np.int64) + 64) // 64)) == {3, 4}        # both word counts
    e = Engine(0, **okw)
    e.set_readset(rs)
    want = g[name + "_res"]
    got = e.window_ed_batch(t)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    assert (want[:, 0] != NOALN).sum() > 300
    # all four band widths in one call: every task is served by the launch of its width
    t1 = ed_tasks(name, n_reads=6, seed=5, wide=False); t2 = ed_tasks(name, n_reads=6, seed=5, wide=True)
    o = scenario_oracle(name)
    mix = np.concatenate([t1, t, t2]); wmix = np.concatenate([o.window_ed(t1), want, o.window_ed(t2)])
    perm = np.random.default_rng(3).permutation(mix.shape[0])
    got = e.window_ed_batch(mix[perm])
    assert (got == wmix[perm]).all()
    bad = t[:1].copy(); bad[0, 8] = 128                         # a band of 257 diagonals: not built
    with pytest.raises(HaoError):
        e.window_ed_batch(bad)
    e.close()


@pytest.mark.parametrize("mode,gen,tk,rk", [(0, ed_global_tasks, "g", "g"), (1, ed_ext_tasks, "x", "x1"), (2, ed_ext_tasks, "x", "x2"), (3, ed_semi_trace_tasks, "s", "s")])
@pytest.mark.parametrize("name", ["hifi", "ont", "nn"])
def test_wide_bands_trace(name, mode, gen, tk, rk):
    from hifiasm_amd.api import Engine
    g = _wide_fixture()
    rs, okw = scenario_reads(name)
    t = gen(name, n_reads=12, wide=2)
    assert t.shape == g[f"{name}_{tk}tasks"].shape and (t == g[f"{name}_{tk}tasks"]).all(), "the task generator drifted: regenerate the fixture"
    want = g[f"{name}_{rk}res"]; wc = g[f"{name}_{rk}cig"]
    off = np.concatenate([[0], np.cumsum(want[:, 5])]).astype(np.int64)
    e = Engine(0, **okw)
    e.set_readset(rs)
    got, gcig = e.window_trace_batch(t, cap=264, mode=mode)
    e.close()
    assert (got == want).all(), (np.flatnonzero((got != want).any(axis=1))[:10], got[(got != want).any(axis=1)][:3], want[(got != want).any(axis=1)][:3])
    bad = [q for q in range(t.shape[0]) if not (gcig[q, :want[q, 5]] == wc[off[q]:off[q + 1]]).all()]
    assert not bad, bad[:10]
    assert (want[:, 0] != NOALN).sum() > 500
