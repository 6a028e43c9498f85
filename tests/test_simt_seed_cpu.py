"""The seed stage's KERNEL SOURCES on the CPU: hifiasm_amd/csrc/hao_query.cuh / hao_query3.cuh compiled by g++ against the emulated workgroup of
tests/simt/hip/hip_runtime.h (one fiber per work-item; ballots, DPP moves, ds_permute and barriers as rendezvous; divergent cross-lane operations are an error),
launched like hao_batch.hpp launches them, compared with the oracle's restatement of minimizers_qgen0 (anchor.cpp:987-1081).  The same sources are what the
`-m gpu` suite runs on the device; here they run where there is no GPU."""
import ctypes as C
import os
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle
import seed_model
import simt_build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(simt_build.build("seed"))
        _lib.simt_seed_run.restype = C.c_int
        _lib.simt_vocab.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def vocab(op, vals, ao=None):
    vals = np.ascontiguousarray(vals, dtype=np.uint32)
    ao = np.ascontiguousarray(ao if ao is not None else [0, 0], dtype=np.uint32)
    out = np.zeros(max(64, int(ao[-1])), dtype=np.uint64)
    err = C.create_string_buffer(400)
    rc = lib().simt_vocab(C.c_int(op), _p(vals), _p(out), _p(ao), C.c_uint32(ao.size - 1), err, C.c_int(400))
    assert rc == 0, err.value.decode()
    return out


def test_cross_lane_vocabulary():
    """the emulator's DPP / permute / ballot semantics through the helpers the kernels use, against closed forms"""
    rng = np.random.default_rng(11)
    for trial in range(20):
        v = rng.integers(0, 1 << 20, 64).astype(np.uint32)
        assert (vocab(0, v) == np.cumsum(v.astype(np.uint64))).all()                                        # hao_wave_incl_scan_u32: row_shr 1,2,4,8 + row_bcast 15 / 31
        s = (v.astype(np.int64) - (1 << 19)).astype(np.int32)
        assert (vocab(1, s.view(np.uint32)).astype(np.int64) == int(s.max())).all()                          # hao_wave_max_i32 (+ readlane 63)
        assert (vocab(2, v) == np.concatenate([[0xabcd], v[:-1]])).all() and (vocab(3, v) == np.concatenate([v[1:], [0xabcd]])).all()      # wave_shr:1 / wave_shl:1
        key = rng.integers(0, 12 if trial % 2 else 512, 64).astype(np.uint32); act = rng.integers(0, 4, 64) > 0
        want = np.array([sum(1 << j for j in range(64) if act[j] and key[j] == key[i]) for i in range(64)], dtype=np.uint64)
        for op in (4, 5):                                                                                       # hao_match_key<9> (v_bitop3 0x90) and hao_match_bits agree with the definition
            got = vocab(op, key | (act.astype(np.uint32) << 31))
            assert (got[act] == want[act]).all(), (trial, op)
        nk = int(rng.integers(1, 200)); cnt = rng.integers(1, 4 if trial % 3 else 150, nk)
        if trial % 5 == 0:
            cnt[:] = 1
        ao = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        k = vocab(6, np.zeros(64), ao)[:int(ao[-1])]                                                           # hao_seed_locate (ds_permute + ballot) over whole reads
        assert (k == np.searchsorted(ao, np.arange(int(ao[-1])), side="right") - 1).all(), trial


def test_the_emulator_raises_its_alarms():
    """kernels that break the rules the emulation checks must be refused with a message, not answered: a ballot only half a wave reaches, lanes of a wave at two
    different cross-lane operations, one lane at a cross-lane operation while its wave stands at the barrier, a store past the dynamic LDS; and the same shapes written correctly pass"""
    out = np.zeros(128, dtype=np.uint32)
    want = {0: "different cross-lane operations / barriers", 1: "different cross-lane operations", 2: "63 at a barrier", 3: "past the end of the dynamic LDS"}
    for which, text in want.items():
        err = C.create_string_buffer(400)
        rc = lib().simt_selfcheck(C.c_int(which), _p(out), err, C.c_int(400))
        assert rc == 1 and text in err.value.decode(), (which, rc, err.value.decode())
    err = C.create_string_buffer(400)
    assert lib().simt_selfcheck(C.c_int(4), _p(out), err, C.c_int(400)) == 0, err.value.decode()
    assert (out[:64] == 32).all()


def seed_inputs(name):
    rs, _ = scenario_reads(name)
    o = scenario_oracle(name)
    keys, off, pos = o.pt_table()
    st = o.stats()
    wgt = seed_model.weight_table(st["high_occ"], st["low_occ"])
    mzs = [o.sketch(r) for r in range(rs.n)]
    mz_off = np.concatenate([[0], np.cumsum([m.shape[0] for m in mzs])]).astype(np.uint64)
    x = np.concatenate([m[:, 0] for m in mzs]).astype(np.uint64); info = np.concatenate([m[:, 1] for m in mzs]).astype(np.uint64)
    if keys.size:
        idx = np.minimum(np.searchsorted(keys, x), keys.size - 1)
        present = keys[idx] == x
        start = np.where(present, off[idx], 0).astype(np.uint64); cnt = np.where(present, off[idx + 1] - off[idx], 0).astype(np.uint64)
    else:
        start = np.zeros(x.size, dtype=np.uint64); cnt = np.zeros(x.size, dtype=np.uint64)
    assert int(cnt.max(initial=0)) < 4096
    lk = np.concatenate([start | cnt << np.uint64(48), np.zeros(1, dtype=np.uint64)])      # (a Python int in the list would promote the array to float64 and round the packed words)
    assert lk.dtype == np.uint64
    return rs, o, dict(mz_off=mz_off, info=info, lk=lk, wgt=wgt.astype(np.uint32), sinfo=np.ascontiguousarray(pos, dtype=np.uint64), len=np.ascontiguousarray(rs.lengths, dtype=np.uint32))


def run_seed(rs, inp, blocks, mode=0, qcap_force=0, want_hq=True):
    n = rs.n; A = int((inp["lk"] >> np.uint64(48)).sum())
    seg = np.zeros(n + 1, dtype=np.uint64); hits = np.zeros((A + 1, 4), dtype=np.uint32); g_tmp = np.zeros(A + 1, dtype=np.uint64); g_cnt = np.zeros(n + 1, dtype=np.uint64)
    hq = np.zeros(A + 64, dtype=np.uint16); stats = np.zeros(8, dtype=np.uint64); err = C.create_string_buffer(400)
    blocks = np.ascontiguousarray(blocks, dtype=np.uint32)
    sinfo = np.concatenate([np.ascontiguousarray(inp["sinfo"], dtype=np.uint64), np.full(16, 0x5a5a5a5a5a5a5a5a, dtype=np.uint64)])      # (the library keeps 8 records of slack behind the index: the merge kernels read whole 32- / 64-byte blocks)
    rc = lib().simt_seed_run(C.c_uint64(n), _p(inp["mz_off"]), _p(inp["info"]), _p(inp["lk"]), _p(inp["wgt"]), _p(sinfo), _p(inp["len"]), C.c_uint64(n),
                             C.c_int(mode), C.c_uint32(qcap_force), _p(blocks), C.c_uint32(blocks.size), C.c_int(1 if want_hq else 0),
                             _p(seg), _p(hits), C.c_uint64(A), _p(g_tmp), _p(g_cnt), _p(hq), _p(stats), err, C.c_int(400))
    assert rc == 0, err.value.decode()
    return seg, hits, g_tmp, g_cnt, hq, stats


def check_reads(o, inp, blocks, seg, hits, g_tmp, g_cnt, hq):
    n_hits = 0
    for r in blocks:
        want = o.seed_hits(int(r)); s, e = int(seg[r]), int(seg[r + 1])
        assert e - s == want.shape[0], (r, e - s, want.shape)
        got = hits[s:e]
        assert (got == want).all(), (r, np.flatnonzero((got != want).any(axis=1))[:5])
        # the group list: one entry per run of hits with the same target, (target << 32 | first hit of the run, relative to the read)
        tid = got[:, 0] & 0x7fffffff
        first = np.flatnonzero(np.concatenate([[True], tid[1:] != tid[:-1]])) if e > s else np.zeros(0, dtype=np.int64)
        assert int(g_cnt[r]) == first.size, (r, int(g_cnt[r]), first.size)
        assert (g_tmp[s:s + first.size] == (tid[first].astype(np.uint64) << np.uint64(32) | first.astype(np.uint64))).all(), r
        # every hit's query minimizer index (what the wire packer's codes use): that minimizer's position is the hit's self_offset
        m0 = int(inp["mz_off"][r]); q = hq[s:e].astype(np.int64)
        assert (((inp["info"][m0 + q] >> np.uint64(28)) & np.uint64((1 << 27) - 1)).astype(np.uint32) == got[:, 2]).all(), r
        n_hits += e - s
    return n_hits


@pytest.mark.parametrize("name,step,mode", [("hifi", 1, 0), ("rr", 1, 0), ("nn", 1, 0), ("ont", 1, 0), ("edge", 4, 0), ("k40", 2, 0), ("hpc0", 2, 0), ("fz2", 2, 0),
                                            ("hifi", 2, 2), ("ont", 2, 2), ("rr", 3, 2), ("rr_heavy", 40, 0), ("rr_heavy", 70, 2)])
def test_seed_kernels_against_the_oracle(name, step, mode):
    rs, o, inp = seed_inputs(name)
    blocks = np.arange(0, rs.n, step)
    out = run_seed(rs, inp, blocks, mode=mode, qcap_force=64 if mode == 2 else 0)
    n_hits = check_reads(o, inp, blocks, *out[:5])
    st = out[5]
    print(f"[simt seed] {name} mode {mode}: {blocks.size} reads, {n_hits} hits, {int(st[0])} cross-lane operations, {int(st[1])} barriers, {int(st[5])} fiber switches; overflow lists {int(st[2])} / {int(st[3])}")
    assert n_hits > 500


def fabricated_index(n_targets, nq, list_len, seed, run_rate=0.08):
    """a position index written directly (no reads behind it): query read 0 with nq minimizers, each with a list of ~list_len records over random targets in
    (rid, pos) order, some targets several times in a list (runs: what the reversal rule for opposite-strand hits is about), some minimizers without a list"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(3000, 9000, n_targets + 1).astype(np.uint32)
    qpos = np.sort(rng.choice(np.arange(60, 60 + 40 * nq), nq, replace=False))
    keys = (np.arange(nq, dtype=np.uint64) * np.uint64(1000003) + np.uint64(17))
    info = []; off = [0]; recs = []
    for q in range(nq):
        zrev = int(rng.integers(0, 2)); span = int(rng.integers(51, 80))
        info.append(0 | int(qpos[q]) << 28 | zrev << 55 | span << 56)
        if q % 13 == 5:
            off.append(off[-1]); continue                                  # a minimizer whose k-mer is not in the index
        k = int(rng.integers(1, 2 * list_len))
        tids = np.sort(rng.integers(1, n_targets + 1, k))
        dup = rng.random(k) < run_rate
        tids[1:][dup[1:]] = tids[:-1][dup[1:]]; tids = np.sort(tids)         # runs of one target
        pos_ = rng.integers(100, 2900, k); rev_ = rng.integers(0, 2, k)
        order = np.lexsort((pos_, tids))
        for i in order:
            recs.append(int(tids[i]) | int(pos_[i]) << 28 | int(rev_[i]) << 55 | span << 56)
        off.append(off[-1] + k)
    mz = np.stack([keys, np.array(info, dtype=np.uint64)], axis=1)
    return mz, keys, np.array(off, dtype=np.int64), np.array(recs, dtype=np.uint64), lens


@pytest.mark.parametrize("n_targets,nq,list_len,mode", [(150, 120, 40, 0), (600, 200, 50, 0), (600, 160, 50, 2), (2500, 260, 60, 0)])
def test_seed_kernels_with_many_bins(n_targets, nq, list_len, mode):
    """reads that overflow the 512-slot table (more than 224 bins: second launch), the 1024-slot table (more than 736: third launch) and the 2048-slot table
    (more than 1760: the third launch's (target, strand) range rounds) - against the formulation model (tests/seed_model.py, itself checked against the oracle)"""
    mz, keys, off, pos, lens = fabricated_index(n_targets, nq, list_len, seed=n_targets + mode)
    wgt = seed_model.weight_table(60, 8)
    want = seed_model.seed_hits_model(mz, keys, off, pos, lens, wgt, 0)
    n = n_targets + 1
    cnt = (off[1:] - off[:-1]).astype(np.uint64)
    inp = dict(mz_off=np.concatenate([np.zeros(1, dtype=np.uint64), np.full(n, nq, dtype=np.uint64)]), info=np.ascontiguousarray(mz[:, 1]),
               lk=np.concatenate([off[:-1].astype(np.uint64) | cnt << np.uint64(48), np.zeros(1, dtype=np.uint64)]), wgt=wgt.astype(np.uint32), sinfo=pos, len=lens)

    class RS:
        pass
    rs = RS(); rs.n = n
    seg, hits, g_tmp, g_cnt, hq, st = run_seed(rs, inp, [0], mode=mode, qcap_force=64 if mode == 2 else 0)
    s, e = int(seg[0]), int(seg[1])
    bins = np.unique(want[:, 0]).size
    print(f"[simt seed] fabricated {n_targets} targets, mode {mode}: {e - s} hits in {bins} bins; overflow lists {int(st[2])} / {int(st[3])}; {int(st[0])} cross-lane operations")
    assert e - s == want.shape[0] and (hits[s:e] == want).all(), np.flatnonzero((hits[s:e] != want).any(axis=1))[:5]
    tid = want[:, 0] & 0x7fffffff
    first = np.flatnonzero(np.concatenate([[True], tid[1:] != tid[:-1]]))
    assert int(g_cnt[0]) == first.size and (g_tmp[s:s + first.size] == (tid[first].astype(np.uint64) << np.uint64(32) | first.astype(np.uint64))).all()
    assert int(st[2]) == (1 if bins > 224 else 0) and int(st[3]) == (1 if bins > 736 else 0), (bins, st[2:4])


# ---- the list-major kernel (hao_query5.cuh): persistent workgroups, a read's position lists staged in LDS and merged by (target, strand) bin ----
_MERGE_ALL = [("hifi", 1, 12), ("rr", 1, 12), ("nn", 1, 12), ("ont", 1, 12), ("edge", 1, 12), ("k40", 1, 12), ("hpc0", 1, 12), ("fz2", 1, 12), ("rr_heavy", 25, 12), ("hifi", 1, 13), ("rr", 1, 13), ("edge", 1, 13)]
# the default CPU suite runs a selection; HAO_SIMT_FULL=1 runs every combination
_MERGE_DEFAULT = {("hifi", 1, 12), ("rr", 1, 12), ("edge", 1, 12), ("ont", 1, 12), ("nn", 1, 12), ("hifi", 1, 13)}


@pytest.mark.parametrize("name,step,mode", [c for c in _MERGE_ALL if os.environ.get("HAO_SIMT_FULL") or c in _MERGE_DEFAULT])
def test_merge_kernel_against_the_oracle(name, step, mode):
    """modes 12 / 13: persistent workgroups of 512 work-items, three of them for the whole read set (so every workgroup runs many reads through its four-stage
    pipeline), records staged in LDS with 16-bit / 32-bit offsets, eight target ranges per read; SIMT_SEED_WIDE=1: the instance for reads with more than 1024 minimizers"""
    rs, o, inp = seed_inputs(name)
    blocks = np.arange(0, rs.n, step)
    out = run_seed(rs, inp, blocks, mode=mode)
    n_hits = check_reads(o, inp, blocks, *out[:5])
    st = out[5]
    print(f"[simt merge] {name} mode {mode}: {blocks.size} reads, {n_hits} hits, {int(st[0])} cross-lane operations, {int(st[1])} barriers; left to the table kernels {int(st[6])}, their overflow lists {int(st[2])} / {int(st[3])}")
    assert n_hits > 500


_RUNS_ALL = [(150, 120, 40, 12, 0.08), (600, 200, 50, 12, 0.08), (2500, 260, 60, 12, 0.08), (40, 300, 60, 12, 0.5), (12, 500, 30, 12, 0.9), (30, 200, 5, 12, 0.3), (25, 540, 20, 12, 0.6), (300, 400, 9, 12, 0.2),
             (60, 1400, 12, 12, 0.3), (60, 1650, 8, 12, 0.3), (60, 1700, 8, 12, 0.3), (3, 900, 20, 12, 0.9), (40, 300, 60, 13, 0.5), (12, 500, 30, 13, 0.9), (700, 1500, 12, 12, 0.05)]
_RUNS_DEFAULT = {(40, 300, 60, 12, 0.5), (12, 500, 30, 12, 0.9), (30, 200, 5, 12, 0.3), (600, 200, 50, 12, 0.08), (60, 1400, 12, 12, 0.3), (60, 1700, 8, 12, 0.3), (3, 900, 20, 12, 0.9), (12, 500, 30, 13, 0.9)}


@pytest.mark.parametrize("n_targets,nq,list_len,mode,run_rate", [c for c in _RUNS_ALL if os.environ.get("HAO_SIMT_FULL") or c in _RUNS_DEFAULT])
def test_merge_kernel_with_runs(n_targets, nq, list_len, mode, run_rate):
    """fabricated indexes: many targets (many steps with a single hit), and lists in which a target comes several times in a row (the redo of a target with the
    per-row runs, forward strand in list order, opposite strand in reverse list order) - up to lists that are a handful of long runs"""
    mz, keys, off, pos, lens = fabricated_index(n_targets, nq, list_len, seed=7 * n_targets + mode, run_rate=run_rate)
    wgt = seed_model.weight_table(60, 8)
    want = seed_model.seed_hits_model(mz, keys, off, pos, lens, wgt, 0)
    n = n_targets + 1
    cnt = (off[1:] - off[:-1]).astype(np.uint64)
    inp = dict(mz_off=np.concatenate([np.zeros(1, dtype=np.uint64), np.full(n, nq, dtype=np.uint64)]), info=np.ascontiguousarray(mz[:, 1]),
               lk=np.concatenate([off[:-1].astype(np.uint64) | cnt << np.uint64(48), np.zeros(1, dtype=np.uint64)]), wgt=wgt.astype(np.uint32), sinfo=pos, len=lens)

    class RS:
        pass
    rs = RS(); rs.n = n
    seg, hits, g_tmp, g_cnt, hq, st = run_seed(rs, inp, [0], mode=mode)
    s, e = int(seg[0]), int(seg[1])
    print(f"[simt merge] fabricated {n_targets} targets, {nq} minimizers, run rate {run_rate}, mode {mode}: {e - s} hits in {np.unique(want[:, 0]).size} bins; left to the table kernels {int(st[6])}")
    assert e - s == want.shape[0] and (hits[s:e] == want).all(), np.flatnonzero((hits[s:e] != want).any(axis=1))[:5]
    tid = want[:, 0] & 0x7fffffff
    first = np.flatnonzero(np.concatenate([[True], tid[1:] != tid[:-1]]))
    assert int(g_cnt[0]) == first.size and (g_tmp[s:s + first.size] == (tid[first].astype(np.uint64) << np.uint64(32) | first.astype(np.uint64))).all()
    m0 = 0; q = hq[s:e].astype(np.int64)
    assert (((inp["info"][m0 + q] >> np.uint64(28)) & np.uint64((1 << 27) - 1)).astype(np.uint32) == want[:, 2]).all()
    if nq > 1536:      # the list-major kernel leaves a read with more than 1536 minimizers (all of them count) - or with more than 96 targets in one wave's range
        assert int(st[6]) == 1
