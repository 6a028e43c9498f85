"""Closed form of the second-level minimizer thinning (mz1_select_mz_h, sketch.cpp:247-330; mz1_qfw :226-246; mz1_hf_select :194-216) - the specification
of a data-parallel sketch_select kernel, written the way a wave would evaluate it and checked against the oracle by
tests/test_select_model_cpu.py.  The device kernel of round 3 (sketch_select_kernel, one LANE per read) and the oracle replay the reference's
newest-wins state machine; this file states WHAT that machine computes:

  input   the read's candidate list in position order: hash x, filter-table count cnt (0 = not a high-count k-mer), position pos, k-mer ordinal ord;
          tot_l = valid k-mer iterations of the read; read length; sample_dist (500), w = mz_rewin (1000), k.
  runs    maximal stretches of cnt > 0 entries; a run between bounding positions ps / pe (0 / read length at the ends) may be thinned iff
          q = int((pe - ps) / sample_dist + .499) > 0.  No such run: everything is kept.
  order   high-count entries by (cnt, x); every cnt == 0 entry is larger than all of them (and equal to the other cnt == 0 entries).
  windows W_i = { m <= i : ord[m] + w > ord[i] } for every i >= i0, where i0 = the first entry that closes the first full second-level window (the first i with
          ord[i] >= w + k - 1, or whose successor jumps past it, or the last entry of a read long enough) and W_i0 = [0, i0];
          plus the tail windows [s, n - 1] for s = s_last, s_last + 1, ... while ord[s] + w <= tot_l + 1 (s_last = start of W_(n-1)).
  marks   a high-count entry is MARKED iff it attains the minimum (all ties) of at least one of those windows - i.e. iff
          key[j] == max over the windows that contain j of min(key over the window): two range queries (a range-min per window, a range-max of those
          minima per entry), no sequential state.
  result  cnt == 0 entries are always kept; in a run with q > 0 the marked entries are kept - or, if the run has none, its min(16, q) smallest (cnt, x)
          entries with cnt < pe - ps; high-count entries of runs with q == 0 are dropped (once any run has q > 0)."""
import numpy as np

INF = 1 << 62


def _sparse(vals, op):
    """sparse table for idempotent range queries over a list of comparable values"""
    t = [list(vals)]
    j = 1
    while (1 << j) <= len(vals):
        p = t[-1]; h = 1 << (j - 1)
        t.append([op(p[i], p[i + h]) for i in range(len(vals) - (1 << j) + 1)])
        j += 1
    return t


def _query(t, a, b, op):
    """op over vals[a .. b] (inclusive)"""
    j = (b - a + 1).bit_length() - 1
    return op(t[j][a], t[j][b - (1 << j) + 1])


def select_keep(x, cnt, pos, ordv, tot_l, length, sample_dist=500, w=1000, k=51):
    """indices of the candidates that survive mz1_select_mz_h"""
    n = len(x)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    x = [int(v) for v in x]; cnt = [int(v) for v in cnt]; pos = [int(v) for v in pos]; ordv = [int(v) for v in ordv]
    # runs of high-count entries and their sampling quota
    runs, last0 = [], -1
    for i in range(n + 1):
        if i == n or cnt[i] == 0:
            if i - last0 > 1:
                ps = 0 if last0 < 0 else pos[last0]; pe = length if i == n else pos[i]
                runs.append((last0 + 1, i, pe - ps, int((pe - ps) / sample_dist + .499)))
            last0 = i
    if not any(q > 0 for *_, q in runs):
        return np.arange(n)
    ws = w + k - 1
    i0 = next((i for i in range(n) if ordv[i] >= ws or (i + 1 < n and ordv[i] < ws and ordv[i + 1] > ws) or (i + 1 == n and tot_l >= ws and ordv[i] < ws)), None)
    keep = [c == 0 for c in cnt]
    if i0 is None:                      # (the reference then drops every high-count entry: sketch.cpp:326-329 runs regardless)
        return np.flatnonzero(keep)
    key = [(cnt[i], x[i]) if cnt[i] > 0 else (INF, 0) for i in range(n)]
    tmin = _sparse(key, min)
    # start of every window that ends at an entry: two-pointer = each lane's binary search for the first m with ord[m] + w > ord[i]
    start = [0] * n
    s = 0
    for i in range(i0 + 1, n):
        while s < i and ordv[s] + w <= ordv[i]:
            s += 1
        start[i] = s
    # minima of the windows W_i (i >= i0); M[i] for i < i0 is "no window"
    NONE = (-1, -1)
    M = [NONE] * n
    for i in range(i0, n):
        M[i] = _query(tmin, start[i] if i > i0 else 0, i, min)
    tmax = _sparse(M, max)
    # tail windows [s, n - 1]: their minima are suffix minima; entry j is in the tail windows with start <= j
    s_last = start[n - 1] if n - 1 > i0 else 0
    tail_hi = s_last - 1                # last tail start (inclusive); none if the loop never runs
    t = s_last
    while t < n and ordv[t] + w <= tot_l + 1:
        tail_hi = t; t += 1
    suf = [NONE] * (n + 1)              # running max over tail windows of their minima, as a function of the largest start considered
    best = NONE
    for t in range(s_last, tail_hi + 1):
        best = max(best, _query(tmin, t, n - 1, min))
        suf[t] = best
    marked = [False] * n
    for j in range(n):
        if cnt[j] == 0:
            continue
        # windows W_i containing j: i from max(j, i0) to the last i whose window still starts at or before j
        lo = max(j, i0)
        hi = lo - 1
        a, b = lo, n - 1                # last i with start[i] <= j (start is non-decreasing): binary search
        while a <= b:
            m = (a + b) >> 1
            if (start[m] if m > i0 else 0) <= j:
                hi = m; a = m + 1
            else:
                b = m - 1
        cand = _query(tmax, lo, hi, max) if hi >= lo else NONE
        if tail_hi >= s_last and j >= s_last:      # tail windows with start <= j
            cand = max(cand, suf[min(j, tail_hi)])
        marked[j] = cand == key[j]
    for a, b, span, q in runs:
        if q <= 0:
            continue
        idx = [m for m in range(a, b) if marked[m]]
        if idx:
            for m in idx:
                keep[m] = True
        else:                           # mz1_hf_select: the min(16, q) smallest of the run, kept if rarer than the run is long
            for m in sorted(range(a, b), key=lambda m: key[m])[: min(16, q)]:
                if cnt[m] < span:
                    keep[m] = True
    return np.flatnonzero(keep)
