"""The window-alignment kernels' lane functions on the CPU (f3, hifiasm_amd/csrc/hao_align.cuh): tests/ed_model.cpp compiles the very file the device compiles
- text staging, pattern streaming out of the packed reads, the per-column step, final scans, traceback - with g++ and replaces a wave by a loop over 64
lanes; the host flow (tasks sorted by text window, tiles, text segments, column-free first sweep + selection + sliced second sweep) is mirrored too.  Compared
with the oracle (pinned to the reference's own ed_band_cal_* functions by tests/test_oracle_ed.py) on the task sets the GPU tests use: both strands, clipped
patterns, N bases, unrelated pairs, degenerate lengths, one- and two-word bands."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import ed_tasks, ed_global_tasks, ed_semi_trace_tasks, ed_ext_tasks, scenario_reads, scenario_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = [("hifi", False), ("ont", False), ("nn", False), ("edge", False), ("hifi", True), ("ont", True)]
NOALN = 2**31 - 1


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("edmodel") / "libedmodel.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "hifiasm_amd", "csrc"), os.path.join(ROOT, "tests", "ed_model.cpp"), "-o", out])
    L = C.CDLL(out)
    vp = C.c_void_p
    L.hao_model_window.argtypes = [C.c_int, vp, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint32, C.c_uint64, C.c_int]
    return L


def _run(L, rs, tasks, mode, cap=136, slice_bytes=4 << 30, band=0):
    packed = np.concatenate([np.ascontiguousarray(rs.packed, dtype=np.uint8), np.zeros(16, dtype=np.uint8)])      # the store's 16 bytes of slack
    pk_off = np.ascontiguousarray(rs.pk_off, dtype=np.uint64); ln = np.ascontiguousarray(rs.lengths, dtype=np.uint32)
    ns_off = ns = None
    if rs.codes is not None and (rs.codes > 3).any():
        sites, off = [], [0]
        co = rs.code_off.astype(np.int64)
        for r in range(rs.n):
            p = np.flatnonzero(rs.codes[co[r]:co[r + 1]] > 3).astype(np.uint32)
            sites.append(p); off.append(off[-1] + p.size)
        ns = np.concatenate(sites).astype(np.uint32) if off[-1] else np.zeros(1, dtype=np.uint32)
        ns_off = np.array(off, dtype=np.uint64)
    t = np.ascontiguousarray(tasks, dtype=np.uint32)
    n = t.shape[0]
    ed = np.zeros((n, 2), dtype=np.int32); tr = np.zeros((n, 6), dtype=np.int32); cig = np.zeros((n, cap), dtype=np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None      # noqa: E731
    assert L.hao_model_window(mode, p(packed), p(pk_off), p(ln), p(ns_off), p(ns), p(t), n, p(ed), p(tr), p(cig), cap, slice_bytes, band) == 0
    return ed, tr, cig


@pytest.mark.parametrize("name,wide", SETS)
def test_model_window_ed(model, name, wide):
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    t = ed_tasks(name, n_reads=40, seed=5, wide=wide)
    got, _, _ = _run(model, rs, t, 4)
    want = o.window_ed(t)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    assert (want[:, 0] != NOALN).sum() > 300


@pytest.mark.parametrize("mode,gen", [(0, ed_global_tasks), (1, ed_ext_tasks), (2, ed_ext_tasks), (3, ed_semi_trace_tasks)])
@pytest.mark.parametrize("name,wide", SETS)
def test_model_window_trace(model, name, wide, mode, gen):
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    t = gen(name, n_reads=40, seed=7 + mode, wide=wide)
    if not wide and mode == 0:                                   # both band widths in one call
        t = np.concatenate([t, gen(name, n_reads=6, seed=3, wide=True)])
    want, wcig = o.window_trace(t, cap=136, mode=mode)
    for slice_bytes in (4 << 30, 40 << 20):                      # the second sweep in one slice / in several
        _, got, gcig = _run(model, rs, t, mode, cap=136, slice_bytes=slice_bytes)
        assert (got == want).all(), (slice_bytes, np.flatnonzero((got != want).any(axis=1))[:10])
        bad = [q for q in range(t.shape[0]) if not (gcig[q, :want[q, 5]] == wcig[q, :want[q, 5]]).all()]
        assert not bad, (slice_bytes, bad[:10])
    assert (want[:, 0] != NOALN).sum() > 200


# ---- bands of three and four words (thre 64 .. 127): the lane functions over hao_wide<3> / hao_wide<4> against the REFERENCE's own *_infi_* functions ----
# (tests/golden/ed_wide.npz, tests/golden/make_golden_ed_wide.py: ed_band_cal_semi_infi_w_absent_diag, ed_band_cal_global_infi_w_trace,
# ed_band_cal_extension_infi_{0,1}_w_trace, ed_band_cal_semi_infi_w_absent_diag_trace with nword = ceil((2 thre + 1) / 64), as cal_exz_infi calls them, Correct.cpp:14508-14565)
def _wide_fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ed_wide.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", ["hifi", "ont", "nn"])
def test_model_wide_bands_distance(model, name):
    g = _wide_fixture()
    rs, okw = scenario_reads(name)
    t = ed_tasks(name, n_reads=12, wide=2)
    assert t.shape == g[name + "_tasks"].shape and (t == g[name + "_tasks"]).all(), "the task generator drifted: regenerate the fixture"
    assert set(np.unique((2 * t[:, 8].astype(np.int64) + 64) // 64)) == {3, 4}        # both word counts
    got, _, _ = _run(model, rs, t, 4, band=1)
    want = g[name + "_res"]
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    assert (want[:, 0] != NOALN).sum() > 300


@pytest.mark.parametrize("mode,gen,tk,rk", [(0, ed_global_tasks, "g", "g"), (1, ed_ext_tasks, "x", "x1"), (2, ed_ext_tasks, "x", "x2"), (3, ed_semi_trace_tasks, "s", "s")])
@pytest.mark.parametrize("name", ["hifi", "ont", "nn"])
def test_model_wide_bands_trace(model, name, mode, gen, tk, rk):
    g = _wide_fixture()
    rs, okw = scenario_reads(name)
    t = gen(name, n_reads=12, wide=2)
    assert t.shape == g[f"{name}_{tk}tasks"].shape and (t == g[f"{name}_{tk}tasks"]).all(), "the task generator drifted: regenerate the fixture"
    want = g[f"{name}_{rk}res"]; wc = g[f"{name}_{rk}cig"]
    off = np.concatenate([[0], np.cumsum(want[:, 5])]).astype(np.int64)
    for slice_bytes in (4 << 30, 80 << 20):
        _, got, gcig = _run(model, rs, t, mode, cap=200, slice_bytes=slice_bytes, band=1)
        assert (got == want).all(), (slice_bytes, np.flatnonzero((got != want).any(axis=1))[:10], got[(got != want).any(axis=1)][:3], want[(got != want).any(axis=1)][:3])
        bad = [q for q in range(t.shape[0]) if not (gcig[q, :want[q, 5]] == wc[off[q]:off[q + 1]]).all()]
        assert not bad, (slice_bytes, bad[:10])
    assert (want[:, 0] != NOALN).sum() > 500
