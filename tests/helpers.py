"""Shared helpers for the test-suite (tests only)."""
import os
import zlib

import numpy as np

from hifiasm_amd import synth
from scenarios import SCENARIOS, BIG_SCENARIOS, build_reads
import oracle_py

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_CACHE = {}


def scenario_reads(name):
    if name not in _CACHE:
        dkw, okw = (SCENARIOS.get(name) or BIG_SCENARIOS[name])
        _CACHE[name] = (build_reads(dkw), okw)
    return _CACHE[name]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = dict(zip([str(x) for x in d.pop("meta_keys")], [int(x) for x in d.pop("meta_vals")]))
    return d


_ORACLES = {}


def scenario_oracle(name):
    """Oracle with ft_gen + pt_gen already run (cached per session)."""
    if name not in _ORACLES:
        rs, okw = scenario_reads(name)
        o = oracle_py.Oracle(rs.codes, rs.code_off, **okw)
        o.ft_gen()
        o.pt_gen()
        _ORACLES[name] = o
    return _ORACLES[name]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


# ---- result digests (hao_batch_digest, include/hao.h; same definition in oracle/ref_harness.cpp) ----
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def dg_mix(z):
    z = np.asarray(z, dtype=np.uint64).copy()
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30); z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27); z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


def dg_stream(stream, words):
    """sum over i of mix(w_i + GOLD * (i + 1) + stream * SALT)  (mod 2^64)"""
    w = np.ascontiguousarray(words).view(np.uint64).ravel()
    if w.size == 0:
        return np.uint64(0)
    with np.errstate(over="ignore"):
        i = np.arange(1, w.size + 1, dtype=np.uint64)
        t = dg_mix(w + np.uint64(0x9E3779B97F4A7C15) * i + np.uint64(stream) * np.uint64(0xD6E8FEB86659FD93))
        return np.add.reduce(t, dtype=np.uint64)


def digest_result(ol, fc, cl):
    """digest of one read's (ol uint32 [n,12], fc uint64, cl uint32 [m,4])"""
    with np.errstate(over="ignore"):
        return np.uint64(dg_stream(1, np.ascontiguousarray(ol, dtype=np.uint32)) + dg_stream(2, np.ascontiguousarray(fc, dtype=np.uint64))
                         + dg_stream(3, np.ascontiguousarray(cl, dtype=np.uint32)))


def digest_hits(kh):
    return np.uint64(dg_stream(4, np.ascontiguousarray(kh, dtype=np.uint32)))


def fold_digests(d, block=256):
    """per-read digests -> one value per block of reads: sum of mix(d_r + GOLD * (r + 1))"""
    d = np.ascontiguousarray(d, dtype=np.uint64)
    with np.errstate(over="ignore"):
        t = dg_mix(d + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, d.size + 1, dtype=np.uint64))
    nb = (d.size + block - 1) // block
    out = np.zeros(nb, dtype=np.uint64)
    with np.errstate(over="ignore"):
        np.add.at(out, np.arange(d.size) // block, t)
    return out


# ---- f3: window alignment tasks (hao_window_ed_batch / ed_band_cal_semi_64_w_absent_diag) ----
# thresholds by band class (wide = 0 / False: one 64-bit word, 1 / True: two words, 2: three or four words = the reference's *_infi_* functions)
_THRE_W = ([0, 3, 8, 15, 24, 31], [32, 40, 50, 63], [64, 80, 95, 96, 110, 127])      # window / candidate pairs
_THRE_U = ([0, 1, 5, 15, 31], [32, 45, 63], [64, 97, 127])                          # unrelated pairs
_THRE_S = ([0, 2, 7], [33, 63], [65, 127])                                          # a read against itself
def ed_tasks_grid(name, n_reads=24, wl=375, seed=1, wide=False):
    """Window / candidate pairs on the reference's FIXED window grid (Correct.cpp:5645, 5993: windows of WINDOW = 375 query bases starting at multiples of
    WINDOW, Hash_Table.h:9): every overlap h_ec_lchain found for a query read contributes one pair per grid window it covers, so the ~2 x coverage candidates of a
    window share one text - the shape the window-alignment kernels are laid out for (tools/bench_ed.py --grid).  Windows an overlap covers only partly (its two
    ends) are clipped to the overlap, as the reference does.  Same columns as ed_tasks."""
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    rng = np.random.default_rng(seed)
    out = []
    L = rs.lengths.astype(np.int64)
    for r in rng.choice(rs.n, size=min(n_reads, rs.n), replace=False):
        ol = o.lchain(int(r))[0]
        thre = int(rng.choice(_THRE_W[int(wide)]))      # (one threshold per read: the reference derives it from the window length)
        for z in ol:
            xs, xe, yid, ys, ye, yrev = int(z[1]), int(z[2]), int(z[4]), int(z[5]), int(z[6]), int(z[7])
            tl = int(L[yid])
            for g0 in range(xs // wl * wl, xe + 1, wl):
                ws, we = max(g0, xs), min(g0 + wl - 1, xe)
                tn = we + 1 - ws
                p0 = ys + (ws - xs) - thre
                p1 = p0 + tn + 2 * thre
                ad = 0
                if p0 < 0:
                    ad, p0 = min(-p0, 2 * thre), 0
                p1 = min(p1, tl)
                if p1 <= p0 or tn <= 0:
                    continue
                out.append((yid, p0, p1 - p0, yrev, int(r), ws, tn, 0, thre, ad))
    t = np.array(out, dtype=np.uint32)
    if wide:
        ai = t[:, 2].astype(np.int64) - t[:, 6] + t[:, 9]
        t = t[ai <= 64 * ((2 * t[:, 8].astype(np.int64) + 1 + 63) // 64)]
    return t


def ed_tasks(name, n_reads=24, wl=775, seed=1, wide=False):
    """wide = thresholds of 32 .. 63 (bands of two 64-bit words: the reference's 128-bit functions).
    (pattern, text) pairs the way the window alignment forms them (Correct.cpp:3897): 775-base query windows of the overlaps h_ec_lchain found,
    against the target region on the overlap's diagonal padded by thre on both sides, clipped at the read ends (abs_diag = bases clipped at the start);
    plus degenerate and unrelated pairs.  uint32 [n,10]: p_rid, p_pos, p_len, p_rev, t_rid, t_pos, t_len, t_rev, thre, abs_diag."""
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    rng = np.random.default_rng(seed)
    out = []
    L = rs.lengths.astype(np.int64)
    for r in rng.choice(rs.n, size=min(n_reads, rs.n), replace=False):
        ol = o.lchain(int(r))[0]
        for z in ol[:: max(1, ol.shape[0] // 12)]:
            xs, xe, yid, ys, ye, yrev = int(z[1]), int(z[2]), int(z[4]), int(z[5]), int(z[6]), int(z[7])
            tl = int(L[yid])
            for ws in range(xs, xe + 1, wl):
                tn = min(wl, xe + 1 - ws)
                thre = int(rng.choice(_THRE_W[int(wide)]))
                p0 = ys + (ws - xs) - thre + int(rng.integers(-3, 4))
                p1 = p0 + tn + 2 * thre
                ad = 0
                if p0 < 0:
                    ad, p0 = min(-p0, 2 * thre), 0
                p1 = min(p1, tl)
                if p1 <= p0 or tn <= 0:
                    continue
                out.append((yid, p0, p1 - p0, yrev, int(r), ws, tn, 0, thre, ad))
    # unrelated pairs, tiny strings, reverse-strand text, pattern shorter than the text
    for _ in range(300):
        a, b = (int(x) for x in rng.integers(0, rs.n, 2))
        if L[a] < 2 or L[b] < 2:
            continue
        tn = int(rng.integers(1, min(900, int(L[b])) + 1))
        pn = int(rng.integers(1, min(1000, int(L[a])) + 1))
        thre = int(rng.choice(_THRE_U[int(wide)]))
        out.append((a, int(rng.integers(0, L[a] - pn + 1)), pn, int(rng.integers(0, 2)), b, int(rng.integers(0, L[b] - tn + 1)), tn, int(rng.integers(0, 2)),
                    thre, int(rng.integers(0, 2 * thre + 1))))
    t = np.array(out, dtype=np.uint32)
    if wide:      # the final scan reads bit i of VP / VN for i < p_len - t_len + abs_diag: beyond the band's words the reference's multi-word code indexes the neighbouring
        ai = t[:, 2].astype(np.int64) - t[:, 6] + t[:, 9]      # vectors of its bit_extz_t (a one-word band just wraps its shift count): keep what is defined
        t = t[ai <= 64 * ((2 * t[:, 8].astype(np.int64) + 1 + 63) // 64)]
    return t


def ed_global_tasks(name, n_reads=24, wl=775, seed=2, wide=False):
    """(pattern, text) pairs for the GLOBAL window alignment with traceback (ed_band_cal_global_64_w_trace): query windows of the overlaps against the
    target interval on the overlap's diagonal (same length up to a few bases), plus unrelated / tiny / too-different pairs.  Same record layout as
    ed_tasks (abs_diag = 0)."""
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    rng = np.random.default_rng(seed)
    out = []
    L = rs.lengths.astype(np.int64)
    for r in rng.choice(rs.n, size=min(n_reads, rs.n), replace=False):
        ol = o.lchain(int(r))[0]
        for z in ol[:: max(1, ol.shape[0] // 12)]:
            xs, xe, yid, ys, ye, yrev = int(z[1]), int(z[2]), int(z[4]), int(z[5]), int(z[6]), int(z[7])
            tl = int(L[yid])
            for ws in range(xs, xe + 1, wl):
                tn = min(wl, xe + 1 - ws)
                thre = int(rng.choice(_THRE_W[int(wide)]))
                p0 = max(0, ys + (ws - xs) + int(rng.integers(-2, 3)))
                p1 = min(tl, p0 + tn + int(rng.integers(-3, 4)))
                if p1 <= p0 or tn <= 0:
                    continue
                out.append((yid, p0, p1 - p0, yrev, int(r), ws, tn, 0, thre, 0))
    for _ in range(300):
        a, b = (int(x) for x in rng.integers(0, rs.n, 2))
        if L[a] < 2 or L[b] < 2:
            continue
        tn = int(rng.integers(1, min(400, int(L[b])) + 1))
        thre = int(rng.choice(_THRE_U[int(wide)]))
        pn = max(1, min(int(L[a]), tn + int(rng.integers(-thre - 2, thre + 3))))
        out.append((a, int(rng.integers(0, L[a] - pn + 1)), pn, int(rng.integers(0, 2)), b, int(rng.integers(0, L[b] - tn + 1)), tn, int(rng.integers(0, 2)), thre, 0))
    for a in range(min(rs.n, 40)):      # a read against itself and against its neighbourhood: exact and near-exact pairs, short strings
        if L[a] < 40:
            continue
        n_ = int(rng.integers(1, 40)); p_ = int(rng.integers(0, L[a] - n_ + 1)); thre = int(rng.choice(_THRE_S[int(wide)]))
        out.append((a, p_, n_, 0, a, p_, n_, 0, thre, 0))
        out.append((a, p_, n_, 1, a, int(L[a]) - p_ - n_, n_, 1, thre, 0))
        if p_ + n_ + 1 <= L[a]:
            out.append((a, p_, n_ + 1, 0, a, p_, n_, 0, max(1, thre), 0))
    return np.array(out, dtype=np.uint32)


def ed_semi_trace_tasks(name, n_reads=24, seed=3, wide=False):
    """tasks for the semi-global alignment WITH traceback (ed_band_cal_semi_64_w_absent_diag_trace): ed_tasks' pairs whose band covers the pattern
    (0 <= p_len - t_len + abs_diag <= 2 thre, t_len > abs_diag) - outside of that the reference's traceback indexes its column words out of range."""
    t = ed_tasks(name, n_reads=n_reads, seed=seed, wide=wide).astype(np.int64)
    ai = t[:, 2] - t[:, 6] + t[:, 9]
    keep = (ai >= 0) & (ai <= 2 * t[:, 8]) & (t[:, 6] > t[:, 9])
    return t[keep].astype(np.uint32)


def ed_ext_tasks(name, n_reads=24, wl=775, seed=4, wide=False):
    """(pattern, text) pairs for the extension alignments with traceback (ed_band_cal_extension_64_{0,1}_w_trace): both strings start (forward) or end
    (backward) together at a point of an overlap's diagonal; either may be the longer one, by a little or by a lot; plus unrelated and tiny pairs."""
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    rng = np.random.default_rng(seed)
    out = []
    L = rs.lengths.astype(np.int64)
    for r in rng.choice(rs.n, size=min(n_reads, rs.n), replace=False):
        ol = o.lchain(int(r))[0]
        for z in ol[:: max(1, ol.shape[0] // 12)]:
            xs, xe, yid, ys, ye, yrev = int(z[1]), int(z[2]), int(z[4]), int(z[5]), int(z[6]), int(z[7])
            tl = int(L[yid])
            for ws in range(xs, xe + 1, wl):
                tn = min(int(rng.integers(20, wl + 1)), xe + 1 - ws)
                thre = int(rng.choice(_THRE_W[int(wide)]))
                p0 = max(0, ys + (ws - xs) + int(rng.integers(-1, 2)))
                pn = tn + int(rng.choice([-60, -9, -2, 0, 1, 3, 12, 80]))
                p1 = min(tl, p0 + max(1, pn))
                if p1 <= p0 or tn <= 0:
                    continue
                out.append((yid, p0, p1 - p0, yrev, int(r), ws, tn, 0, thre, 0))
    for _ in range(300):
        a, b = (int(x) for x in rng.integers(0, rs.n, 2))
        if L[a] < 2 or L[b] < 2:
            continue
        tn = int(rng.integers(1, min(300, int(L[b])) + 1)); pn = int(rng.integers(1, min(300, int(L[a])) + 1))
        thre = int(rng.choice(_THRE_U[int(wide)]))
        out.append((a, int(rng.integers(0, L[a] - pn + 1)), pn, int(rng.integers(0, 2)), b, int(rng.integers(0, L[b] - tn + 1)), tn, int(rng.integers(0, 2)), thre, 0))
    for a in range(min(rs.n, 40)):      # exact and near-exact short pairs
        if L[a] < 60:
            continue
        n_ = int(rng.integers(1, 50)); p_ = int(rng.integers(0, L[a] - n_ - 8)); thre = int(rng.choice(_THRE_S[int(wide)]))
        out.append((a, p_, n_, 0, a, p_, n_, 0, thre, 0))
        out.append((a, p_, n_ + 5, 0, a, p_, n_, 0, thre, 0))
        out.append((a, p_, n_, 0, a, p_, n_ + 5, 0, thre, 0))
    return np.array(out, dtype=np.uint32)


def ed_tasks_grid_all(lengths, ols, lo=0, wl=375, thre=15):
    """ed_tasks_grid's pairs for EVERY read of a batch, in the order hao_window_ed_grid generates them on the device: (query read, grid window, position in ol->list);
    lengths = the lengths of ALL reads, ols[i] = the final overlap list of read lo + i (uint32 [n, 12], hao_ovlp_t).  Pairs whose band would not cover p_len - t_len + abs_diag inside its words are left out
    (bands of more than one word), as hao_window_ed_batch refuses them."""
    L = np.asarray(lengths).astype(np.int64)
    nword = (2 * thre + 1 + 63) // 64
    out = []
    for i, ol in enumerate(ols):
        for w in range((int(L[lo + i]) + wl - 1) // wl):
            g0 = w * wl
            for z in ol:
                xs, xe, yid, ys, yrev = int(z[1]), int(z[2]), int(z[4]), int(z[5]), int(z[7])
                if xs // wl > w or xe // wl < w:
                    continue
                ws, we = max(g0, xs), min(g0 + wl - 1, xe)
                tn = we + 1 - ws
                p0 = ys + (ws - xs) - thre; p1 = p0 + tn + 2 * thre; ad = 0
                if p0 < 0:
                    ad, p0 = min(-p0, 2 * thre), 0
                p1 = min(p1, int(L[yid]))
                if p1 <= p0 or tn <= 0 or (nword > 1 and (p1 - p0) - tn + ad > 64 * nword):
                    continue
                out.append((yid, p0, p1 - p0, yrev, int(z[0]), ws, tn, 0, thre, ad))
    return np.array(out, dtype=np.uint32).reshape(-1, 10)


def device_mem_info():
    """(free, total) bytes from the HIP runtime libhao.so is linked with (not torch's own copy of it: a second runtime in the process may not get the device)"""
    import ctypes
    so = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}, key=lambda p: ("/torch/" in p, p))
    hip = ctypes.CDLL(so[0]); fr, to = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(to)) == 0
    return fr.value, to.value
