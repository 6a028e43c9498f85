"""Streaming delivery (hao_overlap_batch_async / hao_deliver_wait / hao_unpack_hits): what lands in the pinned host arenas - ol->list, fake cigars
and cl->list through the 4-byte wire format - must be bit-identical to the oracle, for every read, with two batches in flight, with the
verbatim-hit list overflowing (grow-and-repack path), and interleaved with the blocking API."""
import os

import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu


def _same(a, b):
    return all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))


@pytest.mark.parametrize("name", ["hifi", "rr", "ont", "edge", "rr_heavy", "long200k", "len65535", "len65535w", "hifi+arena4", "hifi+qmz_raw", "rr+qmz_raw", "hifi+arena_probe"])
def test_delivered_results_equal_the_oracle(name, monkeypatch):
    from hifiasm_amd.api import Engine
    if name.endswith("+arena4"):      # the arenas allocated by hand (mmap + mbind to the GPU's NUMA node + hipHostRegister): what the engine falls back to when hipHostMalloc's pages are elsewhere
        monkeypatch.setenv("HAO_ARENA_NUMA", "4"); name = name.split("+")[0]
    if name.endswith("+arena_probe"):      # the measured placement of the arenas (a timed copy per NUMA node, the arena moved to the best): forced on these small arenas
        monkeypatch.setenv("HAO_DBG_TEST", "arena_probe=1"); name = name.split("+")[0]
    raw_tables = name.endswith("+qmz_raw")
    if raw_tables:      # the minimizer tables of the wire format as 8-byte pairs (what batches with a read of 65 536 bases or more get: "ont", "long200k") for reads that would get 4 bytes
        monkeypatch.setenv("HAO_DBG_TEST", "qmz_raw=1"); name = name.split("+")[0]
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    cuts = [0, rs.n // 3, rs.n // 3, rs.n // 2 + 1, rs.n]          # includes an empty batch
    pending, bad, n_exc, n_cl, packed_ = None, [], 0, 0, set()

    def check(slot, lo, hi):
        nonlocal n_exc, n_cl
        d = e.deliver_wait(slot)
        assert (d.rid_lo, d.n_reads) == (lo, hi - lo)
        if d.n_cl:
            packed_.add(bool(d.qmz_pos))
        n_exc += d.n_exc; n_cl += d.n_cl
        for r in range(lo, hi):
            if not _same(e.delivered_read(d, r), o.lchain(r)):
                bad.append(r)

    for lo, hi in zip(cuts[:-1], cuts[1:]):
        slot = e.overlap_batch_async(lo, hi)
        if pending:                       # the previous batch is consumed while this one's copy is (possibly still) in flight
            check(*pending)
        pending = (slot, lo, hi)
    check(*pending)
    assert not bad, bad[:10]
    assert n_cl > 0
    assert packed_ == {int(rs.lengths.max()) < 65536 and not raw_tables}, (packed_, int(rs.lengths.max()))      # (these scenarios' seed weights are far below 256)
    # the blocking API afterwards (same engine) still serves the same bytes
    e.overlap_batch(0, rs.n)
    for r in range(0, rs.n, 7):
        assert _same(e.h_ec_lchain(r), o.lchain(r)), r
    e.close()
    print(f"[stream] {name}: {n_cl} chained hits, {n_exc} verbatim ({100.0 * n_exc / max(1, n_cl):.3f} %)")


def test_exception_list_overflow_repacks():
    from hifiasm_amd.api import Engine
    os.environ["HAO_DBG_TEST"] = "exc_cap=3,exc_every=5"           # every fifth hit of a chain travels verbatim
    try:
        rs, okw = scenario_reads("ont")
        o = scenario_oracle("ont")
        e = Engine(0, **okw)
        e.set_readset(rs)
        e.ha_ft_gen(); e.ha_pt_gen()
        d = e.deliver_wait(e.overlap_batch_async(0, rs.n))
        assert d.n_exc > d.n_cl // 8                # far more verbatim hits than the 3 the list started with: it had to grow and the chains were packed again
        for r in range(rs.n):
            assert _same(e.delivered_read(d, r), o.lchain(r)), r
        e.close()
    finally:
        del os.environ["HAO_DBG_TEST"]


@pytest.mark.parametrize("name", ["ont", "rr"])
def test_cigars_that_travel_raw(name, monkeypatch):
    """a fake cigar with a step the packed word cannot hold travels raw behind the packed ones (bit 63 of its offset): HAO_DBG_TEST=fc_raw_every=3 sends every third
    overlap that way - long ONT cigars (wave-cooperative copy) and short ones alike -, the decoder must give back the same entries"""
    from hifiasm_amd.api import Engine
    import ctypes as C
    import numpy as np
    monkeypatch.setenv("HAO_DBG_TEST", "fc_raw_every=3")
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    d = e.deliver_wait(e.overlap_batch_async(0, rs.n))
    off = np.ctypeslib.as_array(C.cast(d.fc_off, C.POINTER(C.c_uint64)), shape=(int(d.n_ol) + 1,))
    raw = int((off[:-1] >> np.uint64(63)).sum())
    assert abs(raw - int(d.n_ol) // 3) <= 1 and raw > 50
    for r in range(rs.n):
        assert _same(e.delivered_read(d, r), o.lchain(r)), r
    e.close()


def test_ol_only_delivery():
    """the final round needs ol->list only (h_ec_lchain_fast_new reads no chained hits, ecovlp.cpp:5047): cl->list is neither packed nor copied"""
    from hifiasm_amd.api import Engine, DELIVER_OL
    rs, okw = scenario_reads("hifi")
    o = scenario_oracle("hifi")
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    d = e.deliver_wait(e.overlap_batch_async(0, rs.n, parts=DELIVER_OL))
    assert d.n_cl == 0 and d.n_ol > 0 and d.bytes < 200 * d.n_ol
    import ctypes as C
    from hifiasm_amd.api import _arr
    off = _arr(d.ol_off, rs.n + 1, np.uint64)
    for r in range(rs.n):
        m = int(off[r + 1] - off[r]); ol = np.zeros((m, 12), dtype=np.uint32)
        assert e.L.hao_unpack_overlaps(C.byref(d), r, ol.ctypes.data_as(C.c_void_p), m) == m      # (32 bytes per overlap on the wire: hao_ovlp_wire_t)
        assert (ol == o.lchain(r)[0]).all(), r
    e.close()
