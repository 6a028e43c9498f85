"""Executable model of the algorithm of sketch_unit_kernel (hifiasm_amd/csrc/hao_sketch3.cuh) - TEST INFRASTRUCTURE.

The kernel's decisions, restated lane by lane in numpy so that the arithmetic (blocked layout of 16 window entries per lane, 32-bit
proxies, lane-window index rules, candidate verification, the first-window quirk) can be checked on the CPU against the oracle
(tests/test_sketch_model_cpu.py) - the GPU kernel is a transcription of exactly these formulas.

Closed form being evaluated (hao_sketch.cuh header): ordinal j is emitted iff its key equals the minimum key of at
least one valid window of W consecutive ordinals that contains it.  Split in two:

  1. CANDIDATES from 32-bit proxies.  p = a monotone (non-strict) 32-bit image of the key order.  With m'(t) = min p over window t and
     v(j) = max over the valid windows t containing j of m'(t), every true minimum satisfies p(j) == v(j); the converse fails only where
     two different keys share a proxy.  So {p == v} is a superset of the answer, computed with 32-bit min / max only.
  2. VERIFICATION on the sparse candidate list with the full 64-bit keys: the minimum of a window over all keys equals its minimum over the
     candidates in it (its true minima are candidates), so j is emitted iff some valid window t in [j, j+W-1] has no candidate with a
     strictly smaller key: with l = the nearest candidate with a smaller key on the left (within W-1), r = on the right,
        max(j, tmin, l + W) <= min(j + W - 1, tmax, r - 1).
"""
import numpy as np

K, W, E, NL = 51, 51, 16, 64
NENT = E * NL
MW = NENT - 2 * (W - 1)
M64 = (1 << 64) - 1
DUMMY_C = (1 << 28) - 1
INF32 = 0xFFFFFFFF


def hash64(key):
    key = (~key + (key << 21)) & M64
    key ^= key >> 24
    key = (key + (key << 3) + (key << 8)) & M64
    key ^= key >> 14
    key = (key + (key << 2) + (key << 4)) & M64
    key ^= key >> 28
    key = (key + (key << 31)) & M64
    return key


def runs_of(codes, hpc=True):
    """codes (0..3) -> code[1..T], end1[0..T] (end1[e] = 1 + index of the last base of run e; end1[0] = 0)"""
    n = len(codes)
    code, end1 = [0], [0]
    for i in range(n):
        if (not hpc) or i + 1 == n or codes[i + 1] != codes[i]:
            code.append(int(codes[i])); end1.append(i + 1)
    return code, end1


def brev(v, nbits):
    return int(format(v, f"0{nbits}b")[::-1], 2)


class Read:
    def __init__(self, codes, hpc=True, ft_cnt=None):
        self.code, self.end1 = runs_of(codes, hpc)
        self.T = len(self.code) - 1
        self.ft_cnt = ft_cnt
        self._cache = {}

    def key(self, t):
        """ordinal t (k-mer = runs t-K+1..t) -> (c, x, rev, span); dummy: c = DUMMY_C, x = 2^64-1"""
        if t in self._cache:
            return self._cache[t]
        span = self.end1[t] - self.end1[t - K]
        w0 = w1 = 0
        for j in range(K):                      # bit j <-> run t-K+1+j
            c = self.code[t - K + 1 + j]
            w0 |= (c & 1) << j; w1 |= (c >> 1) << j
        mask = (1 << K) - 1
        f0, f1 = brev(w0, K), brev(w1, K)
        r0, r1 = ~w0 & mask, ~w1 & mask
        if f1 < r1:
            x, rev = (hash64(f0) + hash64(f1)) & M64, 0
        else:
            x, rev = (hash64(r0) + hash64(r1)) & M64, 1
        c = 0
        res = None
        if span >= 256:
            res = (DUMMY_C, M64, rev, span)
        else:
            if self.ft_cnt is not None:
                c = self.ft_cnt(x)
                if c >= (1 << 28):
                    res = (DUMMY_C, M64, rev, span)
            if res is None:
                res = (c, x, rev, span)
        self._cache[t] = res
        return res


def proxy(c, x, has_ft, bits=32):
    """monotone 32-bit image of (c, x); bits < 32 coarsens it (tests: more false candidates for the verification to reject)"""
    if x == M64:
        return INF32
    if has_ft:
        p = (0x80000000 | (min(c, 0x7fff) << 16) | (x >> 48)) if c > 0 else (x >> 33)
    else:
        p = x >> 32
    if bits < 32:
        p = (p >> (32 - bits)) << (32 - bits)
    return p


def lt(a, b):
    return (a[0], a[1]) < (b[0], b[1])


def unit_marks(rd, u, has_ft, bits=32):
    """marks of unit u of a read: list of (t, c, x, rev, span) in ordinal order"""
    T = rd.T
    jw0 = K + u * MW
    jw1 = min(jw0 + MW, T + 1)
    if jw0 > T:
        return []
    kw0 = max(K, jw0 - (W - 1))
    kk1 = min(T, jw1 - 1 + W - 1)
    tm0 = max(jw0, W + K - 1)
    keys = [rd.key(kw0 + q) if kw0 + q <= kk1 else (DUMMY_C, M64, 0, 0) for q in range(NENT)]
    p = np.array([proxy(k[0], k[1], has_ft, bits) for k in keys], dtype=np.uint64).reshape(NL, E)
    # ---- sliding minimum over the W entries ending at each entry: own prefix + whole lanes + a suffix of one farther lane ----
    pre = np.minimum.accumulate(p, axis=1)
    suf = np.minimum.accumulate(p[:, ::-1], axis=1)[:, ::-1]
    tot = pre[:, E - 1]
    sh = lambda a, d, fill: np.concatenate([np.full(d, fill, dtype=a.dtype), a[:-d]]) if d > 0 else a      # value of lane L-d  # noqa: E731
    t1, t2, t3 = sh(tot, 1, INF32), sh(tot, 2, INF32), sh(tot, 3, INF32)
    A2 = np.minimum(t1, t2); A3 = np.minimum(A2, t3)
    m = np.zeros_like(p)
    for i in range(E):
        rem = W - 1 - i; nfull = rem // E; part = rem % E
        v = pre[:, i].copy()
        assert nfull in (2, 3)
        v = np.minimum(v, A3 if nfull == 3 else A2)
        if part > 0:
            v = np.minimum(v, sh(suf[:, E - part], nfull + 1, INF32))
        m[:, i] = v
    q = np.arange(NENT).reshape(NL, E)
    t = kw0 + q
    m[(q < W - 1) | (t > kk1) | (t < tm0)] = 0
    # ---- sliding maximum over the W windows that start at each entry ----
    prem = np.maximum.accumulate(m, axis=1)
    sufm = np.maximum.accumulate(m[:, ::-1], axis=1)[:, ::-1]
    totm = prem[:, E - 1]
    shl = lambda a, d: np.concatenate([a[d:], np.zeros(d, dtype=a.dtype)]) if d > 0 else a                  # value of lane L+d  # noqa: E731
    n1, n2, n3 = shl(totm, 1), shl(totm, 2), shl(totm, 3)
    B2 = np.maximum(n1, n2); B3 = np.maximum(B2, n3)
    v = np.zeros_like(p)
    for i in range(E):
        rem = W - (E - i); nfull = rem // E; part = rem % E
        x = sufm[:, i].copy()
        assert nfull in (2, 3)
        x = np.maximum(x, B3 if nfull == 3 else B2)
        if part > 0:
            x = np.maximum(x, shl(prem[:, part - 1], nfull + 1))
        v[:, i] = x
    cand = (p == v).reshape(-1)
    # ---- first-window quirk / short reads (unit 0) ----
    patch_on, prev, pk = 0, -1, None
    force = {}
    if u == 0:
        big = (DUMMY_C, M64)
        if T >= W + K - 1:
            t0 = W + K - 1; pkk = big
            for tt in range(K, t0):
                o = rd.key(tt)
                if not lt(pkk, o):
                    pkk = o[:2]; prev = tt
            if prev >= 0 and pkk[1] != M64:
                o = rd.key(t0)
                if not lt(pkk, o):
                    patch_on = 1; pk = pkk
        else:
            pkk = big
            for tt in range(max(K, T - W + 1), T + 1):
                o = rd.key(tt)
                if not lt(pkk, o):
                    pkk = o[:2]; prev = tt
            patch_on = 2
            if not (prev >= 0 and pkk[1] != M64):
                prev = -1
        if patch_on == 1:
            for tt in range(K, W + K - 1):
                if tt == prev:
                    force[tt] = False; cand[tt - kw0] = True
                elif rd.key(tt)[:2] == pk:
                    force[tt] = True; cand[tt - kw0] = True
        elif patch_on == 2:
            cand[:] = False
            if prev >= 0:
                cand[prev - kw0] = True
    # ---- verification on the candidate list ----
    cl = [int(qq) for qq in np.flatnonzero(cand)]
    out = []
    for ci, qq in enumerate(cl):
        tj = kw0 + qq; kj = keys[qq]
        if kj[1] == M64 or not (jw0 <= tj < jw1):
            continue
        if patch_on == 2:
            ok = tj == prev
        else:
            lmax, rmin = -(1 << 30), 1 << 30
            a = ci - 1
            while a >= 0 and cl[a] >= qq - (W - 1):
                if lt(keys[cl[a]], kj):
                    lmax = kw0 + cl[a]; break
                a -= 1
            a = ci + 1
            while a < len(cl) and cl[a] <= qq + (W - 1):
                if lt(keys[cl[a]], kj):
                    rmin = kw0 + cl[a]; break
                a += 1
            ok = max(tj, tm0, lmax + W) <= min(tj + W - 1, kk1, rmin - 1)
            if tj in force:
                ok = force[tj]
        if ok:
            out.append((tj, kj[0], kj[1], kj[2], kj[3]))
    return out


def sketch_read(codes, hpc=True, ft_cnt=None, bits=32):
    """-> uint64 [n, 4]: x, pos, rev | span << 8, count (ordinal order = position order)"""
    rd = Read(codes, hpc, ft_cnt)
    has_ft = ft_cnt is not None
    res = []
    n_units = max(1, -(-(rd.T - K + 1) // MW)) if rd.T >= K else 0
    for u in range(n_units):
        for (t, c, x, rev, span) in unit_marks(rd, u, has_ft, bits):
            res.append((x, rd.end1[t] - 1, rev | span << 8, c))
    return np.array(res, dtype=np.uint64).reshape(-1, 4)
