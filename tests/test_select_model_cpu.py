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


# ---- the wave-parallel kernel's own element functions (hifiasm_amd/csrc/hao_select2.cuh) through tests/sel2_model.cpp ----
@pytest.fixture(scope="module")
def sel2(tmp_path_factory):
    import ctypes as C, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path_factory.mktemp("sel2") / "libsel2.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-I" + os.path.join(root, "include"),
                           "-I" + os.path.join(root, "hifiasm_amd", "csrc"), os.path.join(root, "tests", "sel2_model.cpp"), "-o", out])
    L = C.CDLL(out)
    L.hao_sel2_model.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
    return L


@pytest.mark.parametrize("name,min_active", [("rr", 20), ("long_rr", 0), ("rr_big", 100), ("fz3", 10), ("nn", 0), ("rr_heavy", 0)])
def test_select2_kernel_model(sel2, name, min_active):
    import ctypes as C
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    active = fallback = 0
    why = {-1: 0, -3: 0, -4: 0}
    for r in range(rs.n):
        mz, x, cnt, pos, od, tot_l = o.sketch_pre(r)
        n = x.size
        if n == 0:
            continue
        info = (cnt.astype(np.uint64) & np.uint64(0xfffffff)) | (pos.astype(np.uint64) << np.uint64(28))
        xx = x.copy(); oo = od.astype(np.uint32); kept = np.zeros(n, dtype=np.int32)
        m = sel2.hao_sel2_model(xx.ctypes.data, info.ctypes.data, oo.ctypes.data, n, int(rs.lengths[r]), int(tot_l), o.opt.sample_dist, o.opt.rewin, o.opt.k, kept.ctypes.data)
        if m == -2:                                   # no high-count candidate: nothing to do
            assert mz.shape[0] == n
            continue
        if m in why:                                  # outside the closed form's reach: the kernel runs the sequential routine
            fallback += 1; why[m] += 1
            continue
        kk = kept[:m]
        active += int(m != n)
        assert m == mz.shape[0], (name, r, m, mz.shape[0])
        assert (x[kk] == mz[:, 0]).all() and (pos[kk] == ((mz[:, 1] >> np.uint64(28)) & np.uint64(0x7ffffff))).all(), (name, r)
    print(f"[select2 model] {name}: {rs.n} reads, {active} thinned, {fallback} left to the sequential routine {why}")
    assert active >= min_active
