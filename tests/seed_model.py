"""Executable statement of the seed stage's formulation (hifiasm_amd/csrc/hao_query.cuh) - what `seed_bin_kernel` computes, lane by lane where it matters - so that
the claim behind it can be checked on the CPU against the oracle's restatement of minimizers_qgen0 (anchor.cpp:987-1081, which materialises 24-byte anchors and
sorts them by (target, strand, query position), then by target offset):

  * an index list is ordered by (rid, pos); walking a read's anchors in GENERATION order - query minimizer q, then list order j - and partitioning them STABLY by
    (target, strand) gives the reference's order, except that opposite-strand hits of one k-mer in one target come out by DEscending target position (inside
    the run of list entries with that target, the anchor at reverse position k takes the record of reverse entry R - 1 - k);
  * the minimizer that holds anchor x of the read comes from the boundary mask of its 64-anchor window (`hao_seed_locate`): only the minimizers that have
    anchors are staged, so their first-anchor offsets increase strictly; lane i looks at boundary kc + 1 + i, a flag travels to the lane of the boundary's
    window position (ds_permute: lanes nobody writes to read 0; boundaries beyond the window park on lane 0), the ballot of the flags is the mask, and a lane's
    minimizer is kc + the number of boundaries at or before it.

`seed_hits_model` returns uint32 [n, 4] rows (readID | strand << 31, offset, self_offset, cnt) like Oracle.seed_hits."""
import numpy as np

MASK28, MASK27 = (1 << 28) - 1, (1 << 27) - 1


def info_fields(v):
    v = int(v)
    return v & MASK28, (v >> 28) & MASK27, (v >> 55) & 1, v >> 56      # rid, pos, rev, span


def weight_table(high_occ, low_occ):
    """hao_seed_weight_table (hao_host.hpp): k_mer_hit::cnt >> 8 for a minimizer whose key occurs n times (anchor.cpp:1065-1076)"""
    hi, lo = max(2, high_occ), max(2, low_occ)
    tab = np.zeros(4096, dtype=np.uint32)
    for n in range(4096):
        if lo < n < hi:
            w = 1
        elif n <= lo:
            w = 2
        else:
            w = int(float(1 + (n + (hi << 1) - 1) // (hi << 1)) ** 1.1)
        tab[n] = min(w, 0xffffff)
    return tab


def locate_window(ao, nk, kc, x0):
    """hao_seed_locate for one 64-anchor window starting at anchor x0; kc = staged minimizer that holds x0.  -> (k of every lane, kc of the next window)"""
    lanes = np.arange(64)
    cand = np.minimum(kc + 1 + lanes, nk)
    p = ao[cand].astype(np.int64) - x0                                   # >= 1 for every real boundary
    got = np.zeros(64, dtype=np.int64)                                   # ds_permute: push a flag to lane p (lanes nobody writes to read 0) ...
    dest = np.where((p >= 0) & (p < 64), p, 0)                           # ... boundaries beyond the window are parked on lane 0
    got[dest] = 1
    got[0] = 0                                                           # lane 0's own anchor is never a boundary
    m = got.cumsum()                                                     # boundaries at or before the lane = popcount(mask & lanes <= me)
    return kc + m, kc + int(m[-1]) + int((p == 64).any())


def seed_hits_model(mz, keys, off, pos, lens, wgt, rid):
    """mz: the read's minimizers uint64 [n,2] (x, info); (keys, off, pos): the position index (sorted keys, CSR offsets, 8-byte records in (rid, pos) order);
    lens: read lengths; wgt: weight_table(...)"""
    # Q1: every minimizer's list (start, count) and the two words its hits share
    idx = np.searchsorted(keys, mz[:, 0]) if keys.size else np.zeros(mz.shape[0], dtype=np.int64)
    present = (idx < keys.size) & (keys[np.minimum(idx, max(0, keys.size - 1))] == mz[:, 0]) if keys.size else np.zeros(mz.shape[0], dtype=bool)
    start = np.where(present, off[np.minimum(idx, keys.size - 1)] if keys.size else 0, 0).astype(np.int64)
    cnt = np.where(present, (off[np.minimum(idx + 1, keys.size)] - off[np.minimum(idx, keys.size - 1)]) if keys.size else 0, 0).astype(np.int64)
    # staging: only the minimizers that have anchors; first-anchor offsets (strictly increasing), list start, index in the full list, strand
    ne = np.flatnonzero(cnt > 0)
    nk = ne.size
    ao = np.concatenate([np.cumsum(cnt[ne]) - cnt[ne], [cnt[ne].sum()]]).astype(np.int64)
    n = int(ao[-1])
    if n == 0:
        return np.zeros((0, 4), dtype=np.uint32)
    # anchors in generation order, 64 at a time, their minimizer through the boundary mask
    kq = np.zeros(n, dtype=np.int64)
    kc = 0
    for x0 in range(0, n, 64):
        k, kc = locate_window(ao, nk, kc, x0)
        kq[x0:min(n, x0 + 64)] = k[:min(64, n - x0)]
    assert (ao[kq] <= np.arange(n)).all() and (np.arange(n) < ao[kq + 1]).all()
    out = []
    for x in range(n):
        k = int(kq[x]); q = int(ne[k]); j = x - int(ao[k]); st = int(start[q]); nl = int(cnt[q])
        _, zpos, zrev, zspan = info_fields(mz[q, 1])
        y = int(pos[st + j]); tid, ypos, yrev, yspan = info_fields(y)
        rev = zrev ^ yrev
        if rev:      # descending target position inside the run of entries with this target
            ja = jb = j
            while ja > 0 and info_fields(pos[st + ja - 1])[0] == tid:
                ja -= 1
            while jb + 1 < nl and info_fields(pos[st + jb + 1])[0] == tid:
                jb += 1
            if jb > ja:
                revs = [z for z in range(ja, jb + 1) if info_fields(pos[st + z])[2] != zrev]
                kpos = revs.index(j)
                y = int(pos[st + revs[len(revs) - 1 - kpos]]); _, ypos, yrev, yspan = info_fields(y)
        offset = (int(lens[tid]) - 1 - (ypos + 1 - yspan)) if rev else ypos
        out.append(((tid << 1) | rev, x, tid | (rev << 31), offset & 0xffffffff, zpos, (int(wgt[min(nl, 4095)]) << 8) | min(zspan, 255)))
    out.sort(key=lambda t: (t[0], t[1]))                                  # the STABLE partition by (target, strand): generation order inside a bin
    return np.array([t[2:] for t in out], dtype=np.uint32).reshape(-1, 4)
