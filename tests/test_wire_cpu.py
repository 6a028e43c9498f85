"""The decoder of the delivery wire format (hao_unpack_hits, include/hao.h) is a pure host function of a delivered view: here the view is built by a
Python encoder written from the format's description (one bit per position = seed hit, rank directory of one entry per 256 positions, code bytes, sorted exception list, chain headers
with the position of their first hit) and must decode back to
the k_mer_hits it was made from.  No GPU involved: the device-side encoder is checked against the oracle by tests/test_gpu_stream.py."""
import ctypes as C

import numpy as np
import pytest

from hifiasm_amd import api


def _encode(reads, rid_lo=7, seed=3, fill=True, packed=False):
    """reads: list of (qmz [(self_offset, cnt)], chains [(w0, [(q, offset), ...])]) -> (Delivery, keep-alive list, expected hits per read).
    Positions are indices among the batch's seed hits: every chain is a run of consecutive positions somewhere in the read's range, with positions that
    belong to no chain (random code bytes, some of them flagged, one of them 0xff with a bogus list entry) between the chains; the byte at a chain's
    first position is arbitrary as well.  packed: the minimizer tables as two arrays (16-bit self_offset, 16-bit cnt: hao_delivery_t::qmz_pos / qmz_cnt) instead of
    8-byte pairs."""
    rng = np.random.default_rng(seed)
    ch_off, cl_off, qm_off = [0], [0], [0]
    hdr, qmz, exc, want = [], [], [], []
    byte_at = []                                 # code byte of every position (0x08 = no code byte)
    h = 0

    def filler(n):
        for _ in range(n if fill else 0):
            b = int(rng.choice([0x08, 0x08, 0x17, 0x2a, 0xff]))
            if b == 0xff:
                exc.append((len(byte_at), 9999, (1, 2, 3, 4)))      # an entry nobody may look up
            byte_at.append(b)

    for qt, chains in reads:
        exp = []
        filler(int(rng.integers(0, 70)))
        for w0, hits in chains:
            q0, o0 = hits[0]
            hdr.append((len(hits), w0, q0, o0, len(byte_at)))
            for i, (q, off) in enumerate(hits):
                exp.append((w0, off, qt[q][0], qt[q][1]))
                if i == 0:
                    byte_at.append(int(rng.choice([0x08, 0x31, 0xff])))      # not the chain's: skipped by the decoder
                    if byte_at[-1] == 0xff:
                        exc.append((len(byte_at) - 1, 7777, (5, 6, 7, 8)))
                else:
                    pq, po = hits[i - 1]
                    dq, dd = q - pq, (off - po) - (qt[q][0] - qt[pq][0])
                    if 1 <= dq <= 15 and -8 <= dd <= 7:
                        code = (dq - 1) << 4 | (dd + 8)
                    else:
                        code = 0xff
                        exc.append((len(byte_at), q, (w0 ^ 0x5a5a, off, qt[q][0], qt[q][1])))      # the entry's readID word is the seed stage's, not the chain's
                    byte_at.append(code)
                h += 1
            filler(int(rng.integers(0, 5)))
        want.append(np.array(exp, dtype=np.uint32).reshape(-1, 4))
        qmz.extend(qt)
        ch_off.append(len(hdr)); cl_off.append(h); qm_off.append(len(qmz))
    n_pos = len(byte_at)
    nw = (n_pos + 63) // 64
    by = np.full(nw * 64, 0x08, dtype=np.uint8); by[:n_pos] = byte_at
    fl = (by != 0x08).astype(np.uint8)
    codes = by[fl == 1]
    bits = np.zeros(max(1, nw), dtype=np.uint64)
    for w in range(nw):
        bits[w] = sum(int(fl[64 * w + b]) << b for b in range(64))
    rank64 = np.zeros(nw + 1, dtype=np.uint32)
    rank64[1:] = np.cumsum(fl.reshape(-1, 64).sum(axis=1)) if nw else 0
    rank = rank64[::4].copy()                                    # the directory that travels: code bytes before every 256th position
    a_hdr = np.zeros(max(1, len(hdr)), dtype=[("n_hits", "<u4"), ("w0", "<u4"), ("q0", "<u4"), ("offset", "<u4"), ("pos", "<u8")])
    for i, t in enumerate(hdr):
        a_hdr[i] = t
    a_qmz = np.array(qmz, dtype=np.uint32).reshape(-1, 2)
    a_codes = np.concatenate([codes, np.zeros(1, dtype=np.uint8)])
    exc.sort(key=lambda e: e[0])
    a_exc = np.zeros(max(1, len(exc)), dtype=[("index", "<u8"), ("q", "<u4"), ("pad", "<u4"), ("hit", "<u4", 4)])
    for i, (idx, q, hit) in enumerate(exc):
        a_exc[i] = (idx, q, 0, hit)
    arrs = [np.array(x, dtype=np.uint64) for x in (ch_off, cl_off, qm_off)] + [a_hdr, a_qmz, bits, rank, a_codes, a_exc]
    d = api.Delivery()
    d.rid_lo, d.n_reads, d.n_chains, d.n_cl, d.n_exc, d.n_codes, d.n_pos = rid_lo, len(reads), len(hdr), h, len(exc), int(codes.size), n_pos
    d.ch_off, d.cl_off, d.qm_off = (a.ctypes.data for a in arrs[:3])
    d.chains, d.qmz, d.cl_bits, d.cl_rank, d.cl_codes, d.cl_exc = (a.ctypes.data for a in arrs[3:])
    if packed:
        assert a_qmz[:, 0].max() < 65536 and a_qmz[:, 1].max() < 65536
        arrs += [a_qmz[:, 0].astype(np.uint16), a_qmz[:, 1].astype(np.uint16)]
        d.qmz, d.qmz_pos, d.qmz_cnt = None, arrs[-2].ctypes.data, arrs[-1].ctypes.data
    return d, arrs, want


@pytest.mark.parametrize("packed", [False, True])
def test_decoder_inverts_the_format(packed):
    rng = np.random.default_rng(5)
    reads = []
    for r in range(9):
        nq = int(rng.integers(40, 400))
        pos = np.cumsum(rng.integers(1, 90, nq)).astype(np.int64) + 50
        qt = [(int(p), int(rng.integers(1, 5)) << 8 | int(rng.integers(20, 120))) for p in pos]
        chains = []
        for c in range(int(rng.integers(0, 7))):
            n = int(rng.integers(1, 150)); q = int(rng.integers(0, 10)); off = int(rng.integers(1000, 50000)); hits = []
            for i in range(n):
                if q >= nq:
                    break
                hits.append((q, off))
                step = 1 if rng.random() < 0.85 else int(rng.integers(2, 22))            # beyond 15 minimizers: an exception
                nq_ = q + step
                if nq_ < nq:
                    shift = 0 if rng.random() < 0.8 else int(rng.integers(-12, 12))      # beyond -8 .. 7: an exception
                    off = off + (qt[nq_][0] - qt[q][0]) + shift
                q = nq_
            if hits:
                chains.append((int(rng.integers(0, 1 << 31)) | (int(rng.integers(0, 2)) << 31), hits))
        reads.append((qt, chains))
    reads.append(([(10, 300)], []))                                                 # a read without chains
    d, keep, want = _encode(reads, packed=packed)
    L = api.lib()
    assert d.n_exc > 0 and d.n_codes > d.n_exc
    for r, exp in enumerate(want):
        out = np.zeros((exp.shape[0] + 1, 4), dtype=np.uint32)
        assert L.hao_unpack_hits(C.byref(d), d.rid_lo + r, out.ctypes.data_as(C.c_void_p), 0) == exp.shape[0]      # cap too small: the count only
        got = L.hao_unpack_hits(C.byref(d), d.rid_lo + r, out.ctypes.data_as(C.c_void_p), exp.shape[0])
        assert got == exp.shape[0] and (out[:got] == exp).all(), f"read {r}"
    assert L.hao_unpack_hits(C.byref(d), d.rid_lo + len(want), None, 0) == 0                                    # not a read of the batch


def test_decoder_edges_of_the_position_space():
    """a one-hit chain at the very last position of a batch whose position count is a multiple of 64 (the decoder must not look at word n_pos / 64), a two-hit chain
    ending there, and first positions that carry a 0xff byte which is not the chain's"""
    qt = [(100 + 37 * i, 1 << 8 | 51) for i in range(200)]
    hit = 0
    for last_len in (1, 2):
        for seed in range(12):
            first = [(q, 9000 + qt[q][0] + (3 if q == 40 else 0)) for q in range(0, 128 - 3 - last_len)]      # 128 positions in all, no filler
            chains = [(5, first), (6 | 1 << 31, [(3, 500), (4, 537), (30, 2000)]), (7, [(q, 4000 + qt[q][0]) for q in range(10, 10 + last_len)])]
            d, keep, want = _encode([(qt, chains)], seed=seed, fill=False)
            assert d.n_pos == 128 and int(keep[3]["pos"][2]) + last_len == d.n_pos and d.n_exc >= 1
            out = np.zeros((want[0].shape[0], 4), dtype=np.uint32)
            got = api.lib().hao_unpack_hits(C.byref(d), d.rid_lo, out.ctypes.data_as(C.c_void_p), want[0].shape[0])
            assert got == want[0].shape[0] and (out == want[0]).all()
            hit += 1
    assert hit == 24


def test_cigar_decoder_inverts_the_format():
    """Fake cigars travel as 4 bytes per entry after an overlap's first - site step in bits 0 .. 19, zigzag(shift step) in bits 20 .. 31, from (x_pos_s, 0) - or,
    when a step does not fit, raw (two words per entry, bit 63 of the overlap's offset): an encoder written from that description, hao_unpack_cigar back."""
    rng = np.random.default_rng(9)

    def entry(site, sh):
        return (site << 32) | (((-sh) << 1 | 1) if sh < 0 else (sh << 1))

    cig, ol, words, off = [], [], [], []
    for j in range(400):
        xs = int(rng.integers(0, 5000)); site, sh = xs, 0; es = [entry(xs, 0)]
        big = j % 7 == 3                                                         # a step beyond 2^20 bases or 2047 diagonals somewhere: the overlap travels raw
        for k in range(int(rng.integers(0, 40))):
            site += int(rng.integers(0, 3000)) + ((1 << 20) + 5 if big and k == 2 else 0)
            sh += int(rng.integers(-40, 41)) + (3000 if big and k == 4 else 0)
            es.append(entry(site, sh))
        if j % 11 == 5:
            es[0] = entry(xs + 1, 0)                                             # a first entry that is not (x_pos_s, 0): raw as well
        row = np.zeros(8, dtype=np.uint32); row[1] = xs; row[7] = len(es)      # hao_ovlp_wire_t: x_pos_s, fc_len
        ol.append(row); cig.append(es)
        packable = es[0] == entry(xs, 0)
        ws = []
        ps, psh = xs, 0
        for e in es[1:]:
            s_, lo = e >> 32, e & 0xffffffff
            sh_ = -(lo >> 1) if lo & 1 else lo >> 1
            ds, dsh = s_ - ps, sh_ - psh
            if not (0 <= ds < (1 << 20) and -2048 <= dsh <= 2047):
                packable = False
            ws.append((ds & 0xfffff) | ((((dsh << 1) ^ (dsh >> 63)) & 0xfff) << 20))
            ps, psh = s_, sh_
        if packable:
            off.append(len(words)); words.extend(ws)
        else:
            off.append(len(words) | (1 << 63))
            for e in es:
                words.extend([e & 0xffffffff, e >> 32])
    assert sum(1 for o in off if o >> 63) > 50 and sum(1 for o in off if not o >> 63) > 200
    a_ol = np.array(ol, dtype=np.uint32); a_off = np.array(off + [len(words)], dtype=np.uint64); a_w = np.array(words, dtype=np.uint32)
    d = api.Delivery()
    d.n_ol, d.n_fc = len(ol), int(a_w.size)
    d.ol, d.fc_off, d.fc = a_ol.ctypes.data, a_off.ctypes.data, a_w.ctypes.data
    L = api.lib()
    for j, es in enumerate(cig):
        out = np.zeros(len(es) + 2, dtype=np.uint64)
        assert L.hao_unpack_cigar(C.byref(d), j, out.ctypes.data_as(C.c_void_p), len(es)) == len(es)
        assert [int(x) for x in out[:len(es)]] == es and out[len(es)] == 0
        assert L.hao_unpack_cigar(C.byref(d), j, out.ctypes.data_as(C.c_void_p), len(es) - 1) == len(es)      # too small a buffer: the count, nothing written
    assert L.hao_unpack_cigar(C.byref(d), len(cig), None, 0) == 0


def test_overlaps_come_back_from_their_wire_records():
    """hao_unpack_overlaps: the 32-byte wire record (y_id | strand << 31, four positions, shared_seed, first-hit index, fc_len) back into hao_ovlp_t with x_id = the read,
    x_pos_strand = 0, align_length = 0; reads of the batch only; a buffer that is too small gets the count and nothing else"""
    rng = np.random.default_rng(11)
    n_reads, rid_lo = 6, 1000
    cnt = [0, 3, 1, 0, 5, 2]
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint64)
    w = np.zeros((int(off[-1]), 8), dtype=np.uint32)
    w[:, 0] = rng.integers(0, 1 << 28, w.shape[0]) | (rng.integers(0, 2, w.shape[0]).astype(np.uint32) << 31)
    w[:, 1:5] = rng.integers(0, 1 << 27, (w.shape[0], 4))
    w[:, 5] = rng.integers(-5, 1 << 20, w.shape[0]).astype(np.int32).view(np.uint32)
    w[:, 6:8] = rng.integers(0, 1 << 30, (w.shape[0], 2))
    d = api.Delivery()
    d.rid_lo, d.n_reads, d.n_ol = rid_lo, n_reads, int(off[-1])
    d.ol_off, d.ol = off.ctypes.data, w.ctypes.data
    L = api.lib()
    for r in range(n_reads):
        m = cnt[r]; out = np.full((m + 1, 12), 0xdeadbeef, dtype=np.uint32)
        assert L.hao_unpack_overlaps(C.byref(d), rid_lo + r, out.ctypes.data_as(C.c_void_p), m) == m
        ww = w[int(off[r]):int(off[r + 1])]
        exp = np.zeros((m, 12), dtype=np.uint32)
        exp[:, 0] = rid_lo + r; exp[:, 1] = ww[:, 1]; exp[:, 2] = ww[:, 2]; exp[:, 4] = ww[:, 0] & 0x7fffffff; exp[:, 5] = ww[:, 3]; exp[:, 6] = ww[:, 4]; exp[:, 7] = ww[:, 0] >> 31
        exp[:, 8] = ww[:, 5]; exp[:, 10] = ww[:, 6]; exp[:, 11] = ww[:, 7]
        assert (out[:m] == exp).all() and (out[m] == 0xdeadbeef).all(), r
        if m:
            out[:] = 7
            assert L.hao_unpack_overlaps(C.byref(d), rid_lo + r, out.ctypes.data_as(C.c_void_p), m - 1) == m and (out == 7).all()
    assert L.hao_unpack_overlaps(C.byref(d), rid_lo + n_reads, None, 0) == 0 and L.hao_unpack_overlaps(C.byref(d), rid_lo - 1, None, 0) == 0
