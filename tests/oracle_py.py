"""ctypes binding of oracle/liboracle.so - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; the product package (hifiasm_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class Opt(C.Structure):
    _fields_ = [("k", C.c_int), ("w", C.c_int), ("hpc", C.c_int), ("sample_dist", C.c_int), ("rewin", C.c_int),
                ("min_hist_cnt", C.c_int), ("max_kmer_cnt", C.c_int), ("high_factor", C.c_double),
                ("max_n_chain", C.c_int), ("is_ont", C.c_int), ("bf_shift", C.c_int), ("bw_thres", C.c_double), ("hg_size", C.c_longlong)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        vp, u8p, u64p, i64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
        L.hao_or_opt_default.argtypes = [C.POINTER(Opt)]
        L.hao_or_hash64.argtypes = [C.c_uint64]; L.hao_or_hash64.restype = C.c_uint64
        L.hao_or_create.argtypes = [u8p, u64p, C.c_uint64, C.POINTER(Opt)]; L.hao_or_create.restype = vp
        L.hao_or_destroy.argtypes = [vp]
        L.hao_or_ft_gen.argtypes = [vp]; L.hao_or_ft_gen.restype = C.c_int
        L.hao_or_pt_gen.argtypes = [vp]; L.hao_or_pt_gen.restype = C.c_int
        L.hao_or_ft_hist.argtypes = [vp]; L.hao_or_ft_hist.restype = i64p
        L.hao_or_pt_hist.argtypes = [vp]; L.hao_or_pt_hist.restype = i64p
        L.hao_or_ft_table.argtypes = [vp, C.POINTER(u64p), C.POINTER(C.POINTER(C.c_int32))]; L.hao_or_ft_table.restype = C.c_uint64
        L.hao_or_pt_table.argtypes = [vp, C.POINTER(u64p), C.POINTER(u64p), C.POINTER(u64p), u64p]; L.hao_or_pt_table.restype = C.c_uint64
        L.hao_or_stats.argtypes = [vp, i64p]
        L.hao_or_ft_cnt.argtypes = [vp, C.c_uint64]; L.hao_or_ft_cnt.restype = C.c_int32
        L.hao_or_kmer_hashes.argtypes = [u8p, C.c_int64, C.c_int, C.c_int, u64p]; L.hao_or_kmer_hashes.restype = C.c_int64
        L.hao_or_sketch.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(vp)]; L.hao_or_sketch.restype = C.c_int64
        L.hao_or_sketch_seq.argtypes = [vp, u8p, C.c_int64, C.c_uint32, C.c_int, C.c_int, C.POINTER(vp)]; L.hao_or_sketch_seq.restype = C.c_int64
        L.hao_or_seed_hits.argtypes = [vp, C.c_uint64, C.POINTER(vp)]; L.hao_or_seed_hits.restype = C.c_int64
        L.hao_or_lchain.argtypes = [vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i64p]
        L.hao_or_lchain.restype = C.c_int64
        L.hao_or_exact.argtypes = [vp, vp, C.c_int64, u8p]
        L.hao_or_window_ed.argtypes = [vp, vp, C.c_int64, vp]
        L.hao_or_window_trace.argtypes = [vp, vp, C.c_int64, C.c_int, vp, vp, C.c_int64]
        L.hao_or_analyze_count.argtypes = [C.c_int, C.c_int, i64p, C.POINTER(C.c_int)]; L.hao_or_analyze_count.restype = C.c_int
        _LIB = L
    return _LIB


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value)
    return np.frombuffer(buf, dtype=dtype).copy()


class Oracle:
    """CPU restatement of the reference path over one read set (codes 0..3, >=4 = N)."""

    def __init__(self, codes: np.ndarray, off: np.ndarray, **kw):
        self.L = lib()
        self.opt = Opt()
        self.L.hao_or_opt_default(C.byref(self.opt))
        for k, v in kw.items():
            setattr(self.opt, k, v)
        self.codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.off = np.ascontiguousarray(off, dtype=np.uint64)
        self.n_reads = self.off.size - 1
        self.h = self.L.hao_or_create(self.codes.ctypes.data_as(C.POINTER(C.c_uint8)), self.off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      self.n_reads, C.byref(self.opt))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hao_or_destroy(self.h)
            self.h = None

    def ft_gen(self):
        return self.L.hao_or_ft_gen(self.h)

    def pt_gen(self):
        return self.L.hao_or_pt_gen(self.h)

    def stats(self):
        out = (C.c_int64 * 8)()
        self.L.hao_or_stats(self.h, out)
        names = ["ft_peak_hom", "ft_peak_het", "ft_cutoff", "max_n_chain", "hom_cov", "het_cov", "high_occ", "low_occ"]
        return dict(zip(names, [int(x) for x in out]))

    def ft_hist(self):
        return _arr(self.L.hao_or_ft_hist(self.h), 4096, np.int64)

    def pt_hist(self):
        return _arr(self.L.hao_or_pt_hist(self.h), 4096, np.int64)

    def ft_table(self):
        k, v = C.POINTER(C.c_uint64)(), C.POINTER(C.c_int32)()
        n = self.L.hao_or_ft_table(self.h, C.byref(k), C.byref(v))
        return _arr(k, n, np.uint64), _arr(v, n, np.int32)

    def pt_table(self):
        k, o, p = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        npos = C.c_uint64()
        n = self.L.hao_or_pt_table(self.h, C.byref(k), C.byref(o), C.byref(p), C.byref(npos))
        return _arr(k, n, np.uint64), _arr(o, n + 1, np.uint64), _arr(p, npos.value, np.uint64)

    def kmer_hashes(self, rid):
        s = self.codes[int(self.off[rid]):int(self.off[rid + 1])]
        out = np.empty(max(1, s.size), dtype=np.uint64)
        n = self.L.hao_or_kmer_hashes(s.ctypes.data_as(C.POINTER(C.c_uint8)), s.size, self.opt.k, self.opt.hpc,
                                      out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out[:n]

    def sketch(self, rid, use_ft=True, sample_dist=None):
        """-> uint64 [n,2] (x, info)"""
        p = C.c_void_p()
        sd = self.opt.sample_dist if sample_dist is None else sample_dist
        n = self.L.hao_or_sketch(self.h, rid, int(use_ft), sd, C.byref(p))
        return _arr(p.value, 2 * n, np.uint64).reshape(-1, 2)

    def sketch_pre(self, rid, sample_dist=None):
        """the sketch of read rid (uint64 [n,2]) plus the candidate list mz1_select_mz_h received: x, cnt, pos, ord arrays and tot_l"""
        vp = C.c_void_p
        self.L.hao_or_sketch_pre.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]
        self.L.hao_or_sketch_pre.restype = C.c_int64
        out, pn, x, cnt, pos, od, tl = vp(), C.c_int64(), vp(), vp(), vp(), vp(), C.c_int64()
        sd = self.opt.sample_dist if sample_dist is None else sample_dist
        n = self.L.hao_or_sketch_pre(self.h, rid, sd, C.byref(out), C.byref(pn), C.byref(x), C.byref(cnt), C.byref(pos), C.byref(od), C.byref(tl))
        m = pn.value
        return (_arr(out.value, 2 * n, np.uint64).reshape(-1, 2), _arr(x.value, m, np.uint64), _arr(cnt.value, m, np.uint32), _arr(pos.value, m, np.uint32),
                _arr(od.value, m, np.uint64), tl.value)

    def sketch_seq(self, codes, rid=0, use_ft=True, sample_dist=None):
        p = C.c_void_p()
        sd = self.opt.sample_dist if sample_dist is None else sample_dist
        s = np.ascontiguousarray(codes, dtype=np.uint8)
        n = self.L.hao_or_sketch_seq(self.h, s.ctypes.data_as(C.POINTER(C.c_uint8)), s.size, rid, int(use_ft), sd, C.byref(p))
        return _arr(p.value, 2 * n, np.uint64).reshape(-1, 2)

    def seed_hits(self, rid):
        """-> uint32 [n,4] (readID|strand<<31, offset, self_offset, cnt)"""
        p = C.c_void_p()
        n = self.L.hao_or_seed_hits(self.h, rid, C.byref(p))
        return _arr(p.value, 4 * n, np.uint32).reshape(-1, 4)

    def lchain(self, rid):
        """-> (ol uint32 [n,12], fc uint64, fc_off uint64 [n+1], cl uint32 [m,4])"""
        ol, fc, fco, cl = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        cln = C.c_int64()
        n = self.L.hao_or_lchain(self.h, rid, C.byref(ol), C.byref(fc), C.byref(fco), C.byref(cl), C.byref(cln))
        fo = _arr(fco.value, n + 1, np.uint64)
        return (_arr(ol.value, 12 * n, np.uint32).reshape(-1, 12), _arr(fc.value, int(fo[-1]) if n >= 0 and fo.size else 0, np.uint64), fo,
                _arr(cl.value, 4 * cln.value, np.uint32).reshape(-1, 4))


_META_NAMES = ["n_reads", "k", "w", "hom_cov_ft", "ft_peak_hom", "ft_peak_het", "max_n_chain", "hom_cov", "het_cov", "high_occ", "low_occ",
               "n_ft", "n_ptk", "n_ptp", "tot_ol", "tot_cl", "tot_kh", "ft_distinct", "pt_distinct", "is_ont", "sample_dist", "rewin",
               "max_kmer_cnt", "high_factor_x1000"]


def _exact(self, ol):
    """exact-overlap flags (uint8) of overlaps ol (uint32 [n,12], as lchain() returns them)"""
    a = np.ascontiguousarray(ol, dtype=np.uint32)
    out = np.zeros(a.shape[0], dtype=np.uint8)
    if a.shape[0]:
        self.L.hao_or_exact(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


Oracle.exact = _exact


def _window_ed(self, tasks):
    """tasks uint32 [n,10] -> int32 [n,2] (err, pe): ed_band_cal_semi_64_w_absent_diag"""
    t = np.ascontiguousarray(tasks, dtype=np.uint32).reshape(-1, 10)
    out = np.zeros((t.shape[0], 2), dtype=np.int32)
    if t.shape[0]:
        self.L.hao_or_window_ed(self.h, t.ctypes.data_as(C.c_void_p), t.shape[0], out.ctypes.data_as(C.c_void_p))
    return out


Oracle.window_ed = _window_ed


def _window_trace(self, tasks, cap=80, mode=0):
    """tasks uint32 [n,10] -> (int32 [n,6] (err, ps, pe, ts, te, cigar entries), uint16 [n,cap] cigars): mode 0 ed_band_cal_global_64_w_trace,
    mode 3 ed_band_cal_semi_64_w_absent_diag_trace, each + gen_trace"""
    t = np.ascontiguousarray(tasks, dtype=np.uint32).reshape(-1, 10)
    out = np.zeros((t.shape[0], 6), dtype=np.int32); cig = np.zeros((t.shape[0], cap), dtype=np.uint16)
    if t.shape[0]:
        self.L.hao_or_window_trace(self.h, t.ctypes.data_as(C.c_void_p), t.shape[0], mode, out.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p), cap)
    return out, cig


Oracle.window_trace = _window_trace


def load_ref_meta(prefix: str):
    return dict(zip(_META_NAMES, [int(x) for x in np.fromfile(prefix + ".meta.i64", dtype=np.int64)]))


def load_ref_dump(prefix: str):
    """Read a ref_harness --dump PREFIX directory into a dict of numpy arrays."""
    ext = {"u64": np.uint64, "i64": np.int64, "u32": np.uint32, "i32": np.int32, "u8": np.uint8}
    d = {}
    base = os.path.basename(prefix)
    for fn in os.listdir(os.path.dirname(prefix)):
        if not fn.startswith(base + "."):
            continue
        _, name, e = fn.rsplit(".", 2)
        d[name] = np.fromfile(os.path.join(os.path.dirname(prefix), fn), dtype=ext[e])
    m = d["meta"]
    names = ["n_reads", "k", "w", "hom_cov_ft", "ft_peak_hom", "ft_peak_het", "max_n_chain", "hom_cov", "het_cov", "high_occ", "low_occ",
             "n_ft", "n_ptk", "n_ptp", "tot_ol", "tot_cl", "tot_kh", "ft_distinct", "pt_distinct", "is_ont", "sample_dist", "rewin",
             "max_kmer_cnt", "high_factor_x1000"]
    d["meta"] = dict(zip(names, [int(x) for x in m]))
    return d
