"""GPU parity of the sharded (one engine per shard of reads) path.

A 1-GPU box cannot host two RCCL ranks, so the exchange logic (hash-range k-mer counting, all-gather of
minimizers, global read ids) is exercised through the loopback backend: two engines in one process, one host
thread each, exchanging through device copies - same code path above the transport.  The RCCL backend itself is
exercised with a world of one rank.  Every rank's results must equal the oracle's results for its reads."""
import threading

import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu


def _shard(rs, lo, hi):
    from hifiasm_amd.synth import ReadSet
    pk = rs.packed[int(rs.pk_off[lo]):int(rs.pk_off[hi])]
    co = rs.codes[int(rs.code_off[lo]):int(rs.code_off[hi])]
    return ReadSet(lo, rs.lengths[lo:hi].copy(), pk.copy(), (rs.pk_off[lo:hi + 1] - rs.pk_off[lo]).copy(), co.copy(),
                   (rs.code_off[lo:hi + 1] - rs.code_off[lo]).copy())


def _check_rank(e, o, lo, hi, errors):
    for r in range(lo, hi):
        ol, fc, fo, cl = e.h_ec_lchain(r - lo)
        ool, ofc, ofo, ocl = o.lchain(r)
        if not (ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and cl.shape == ocl.shape and (cl == ocl).all()):
            errors.append(f"read {r}")


@pytest.mark.parametrize("name", ["hifi", "rr", "nn", "bf24"])
@pytest.mark.parametrize("world", [2, 3])
def test_loopback_world(name, world):
    _loopback_world(name, world)


@pytest.mark.parametrize("name,world,passes", [("hifi", 2, 3), ("nn", 3, 2), ("bf24", 2, 4), ("rr", 3, 5)])
def test_loopback_world_ft_in_passes(name, world, passes, monkeypatch):
    """sharded ha_ft_gen in hash-range passes: every pass is one partition + all-to-all-v + count of a P-th of every rank's range (exact) / of every rank's sub-tables
    (Bloom); all ranks run the same number of passes; tables, histograms and every read's overlaps as with one pass"""
    monkeypatch.setenv("HAO_FT_PASSES", str(passes))
    monkeypatch.setenv("HAO_DBG_TEST", "ft_chunk_slots=25000")
    _loopback_world(name, world)


def test_replicated_index_beyond_2_32(monkeypatch):
    """The replicated index has no 2^32-record limit (only a rank's hash partition has, through its sort's arrival index): with 2^32 + 999 unused position
    records in front of every rank's copy of the index (HAO_DBG_TEST=ix_pad=N, 34 GB per rank) all list starts - the ones a partition's owner sends back to the
    minimizers' home ranks included - lie beyond 2^32; tables and every read's overlaps must not change."""
    monkeypatch.setenv("HAO_DBG_TEST", "ix_pad=" + str((1 << 32) + 999))
    _loopback_world("rr", 2)


def _loopback_world(name, world):
    from hifiasm_amd.api import Engine, lib
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    cuts = [rs.n * i // world for i in range(world + 1)]
    grp = lib().hao_loop_create(world)
    errors, stats, engines = [], [None] * world, [None] * world

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            e = Engine(0, **okw)
            e.set_readset(_shard(rs, lo, hi))
            e.set_shard(lo, rs.lengths)
            e.dist_init_loopback(grp, rank)
            e.ha_ft_gen()
            e.ha_pt_gen()
            stats[rank] = (e.stats(), e.hist(0), e.hist(1), e.ft_table(), e.pt_table())
            e.overlap_batch(0, hi - lo)
            engines[rank] = e
        except Exception as ex:  # noqa: BLE001
            errors.append(f"rank {rank}: {ex!r}")

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[:5]
    for rank, e in enumerate(engines):          # the oracle object is not thread-safe: compare after the ranks have joined
        _check_rank(e, o, cuts[rank], cuts[rank + 1], errors)
        e.close()
    lib().hao_loop_destroy(grp)
    assert not errors, errors[:5]
    so = o.stats()
    for st, h0, h1, ft, pt in stats:
        assert st == so
        assert (h0 == o.ft_hist()).all() and (h1 == o.pt_hist()).all()
        assert (ft[0] == o.ft_table()[0]).all() and (ft[1] == o.ft_table()[1]).all()
        ok, ooff, opos = o.pt_table()
        assert (pt[0] == ok).all() and (pt[1] == ooff).all() and (pt[2] == opos).all()


def test_rccl_single_rank():
    """the RCCL transport with world = 1: ncclCommInitRank, all-gather-v / all-to-all-v / all-reduce on one rank"""
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads("hifi")
    o = scenario_oracle("hifi")
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.set_shard(0, rs.lengths)
    e.dist_init(Engine.dist_unique_id(), 0, 1)
    e.ha_ft_gen()
    e.ha_pt_gen()
    assert e.stats() == o.stats()
    e.overlap_batch(0, rs.n)
    errors = []
    _check_rank(e, o, 0, rs.n, errors)
    e.close()
    assert not errors, errors[:5]


def test_ranks_fail_together():
    """one rank with a wrong shard layout (its rid_base is off): EVERY rank must return the layout error from the same collective - nobody may
    be left waiting in the next exchange (the loopback barrier would deadlock the test if they did)"""
    from hifiasm_amd.api import Engine, HaoError, lib
    rs, okw = scenario_reads("hifi")
    world = 2
    cuts = [0, rs.n // 2, rs.n]
    grp = lib().hao_loop_create(world)
    res = [None] * world

    def run(rank):
        lo, hi = cuts[rank], cuts[rank + 1]
        e = Engine(0, **okw)
        e.set_readset(_shard(rs, lo, hi))
        # rank 1 claims a base that overlaps rank 0's range; the local length check passes because the lengths array is handed in rotated
        if rank == 1:
            e.set_shard(0, np.concatenate([rs.lengths[lo:hi], rs.lengths[:lo]]))
        else:
            e.set_shard(lo, rs.lengths)
        e.dist_init_loopback(grp, rank)
        try:
            e.ha_ft_gen()
            res[rank] = "ok"
        except HaoError as ex:
            res[rank] = str(ex)
        e.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "a rank is stuck in a collective"
    lib().hao_loop_destroy(grp)
    assert all(r is not None and "contiguous" in r for r in res), res


def test_loopback_world8_configs1():
    """BASELINE.json configs[1] (5 Mb, 10 000 reads) split over a loopback world of 8 ranks: every rank's tables equal the single-device
    engine's, and every read's digest (ol, fake cigars, cl; seed hits) equals the digest the single-device run produced for it.  The
    single-device run itself is pinned to the reference at this size by test_gpu_fullsize.py / the 5 Mb fixtures of test_gpu_fullgold.py."""
    from hifiasm_amd.api import Engine, lib
    from hifiasm_amd.workloads import workload_reads
    from hifiasm_amd.synth import ReadSet
    rs = workload_reads("bacterial5M_hifi30x")
    e1 = Engine(0)
    e1.set_readset(rs)
    e1.ha_ft_gen(); e1.ha_pt_gen()
    e1.overlap_batch(0, rs.n)
    d_ref, k_ref = e1.batch_digest(rs.n)
    tot_ref = e1.batch_totals()
    ref = (e1.stats(), e1.hist(0), e1.hist(1), e1.ft_table(), e1.pt_table())
    e1.close()
    world = 8
    cuts = [rs.n * i // world for i in range(world + 1)]
    grp = lib().hao_loop_create(world)
    errors, out = [], [None] * world

    def run(rank):
        try:
            lo, hi = cuts[rank], cuts[rank + 1]
            sh = ReadSet(lo, rs.lengths[lo:hi].copy(), rs.packed[int(rs.pk_off[lo]):int(rs.pk_off[hi])].copy(),
                         (rs.pk_off[lo:hi + 1] - rs.pk_off[lo]).copy(), None, None)
            e = Engine(0)
            e.set_readset(sh)
            e.set_shard(lo, rs.lengths)
            e.dist_init_loopback(grp, rank)
            e.ha_ft_gen(); e.ha_pt_gen()
            st = (e.stats(), e.hist(0), e.hist(1), e.ft_table(), e.pt_table())
            e.overlap_batch(0, hi - lo)
            out[rank] = (st, e.batch_digest(hi - lo), e.batch_totals())
            e.close()
        except Exception as ex:  # noqa: BLE001
            errors.append(f"rank {rank}: {ex!r}")

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    lib().hao_loop_destroy(grp)
    assert not errors, errors[:5]
    tot = 0
    for rank, (st, (d, k), t) in enumerate(out):
        lo, hi = cuts[rank], cuts[rank + 1]
        assert st[0] == ref[0] and (st[1] == ref[1]).all() and (st[2] == ref[2]).all()
        assert all((a == b).all() for a, b in zip(st[3], ref[3])) and all((a == b).all() for a, b in zip(st[4], ref[4]))
        assert (d == d_ref[lo:hi]).all(), f"rank {rank}: results of reads {lo + np.flatnonzero(d != d_ref[lo:hi])[:5]} differ from the single-device run"
        assert (k == k_ref[lo:hi]).all(), f"rank {rank}: seed hits differ"
        tot += t["overlaps"]
    assert tot == tot_ref["overlaps"]
