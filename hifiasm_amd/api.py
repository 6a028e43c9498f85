"""Host-side mirror of the reference seam over the C ABI of libhao.so (include/hao.h).

The reference's interface for this path is three functions:
``ha_ft_gen`` (htab.cpp:1136), ``ha_pt_gen`` (htab.cpp:1232) and ``h_ec_lchain``
(anchor.cpp:2302) plus the accessors ``ha_ft_cnt`` / ``ha_pt_get`` and the finer
``mz1_ha_sketch``.  :class:`Engine` exposes them with the same names, argument
meaning and (absence of) error returns: failures raise :class:`HaoError`, mirroring
the reference's ``exit(1)``.  This module is plumbing only (ctypes + numpy); all
compute happens in the HIP kernels.  There is no CPU fallback: constructing an
Engine without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class HaoError(RuntimeError):
    pass


class Opt(C.Structure):
    _fields_ = [("k", C.c_int32), ("w", C.c_int32), ("hpc", C.c_int32), ("sample_dist", C.c_int32), ("rewin", C.c_int32),
                ("min_hist_cnt", C.c_int32), ("max_kmer_cnt", C.c_int32), ("max_n_chain", C.c_int32),
                ("high_factor", C.c_double), ("is_ont", C.c_int32), ("bf_shift", C.c_int32), ("hg_size", C.c_int64)]


class Pass(C.Structure):
    """hao_pass_t: the per-pass arguments of h_ec_lchain (anchor.cpp:2302)"""
    _fields_ = [("bw_thres", C.c_double), ("max_n_chain", C.c_int32), ("high_occ", C.c_uint32), ("low_occ", C.c_uint32),
                ("apend_be", C.c_int32), ("is_accurate", C.c_int32), ("gen_off", C.c_int32), ("mcopy_num", C.c_int32),
                ("mcopy_rate", C.c_double), ("chain_cutoff", C.c_uint32), ("mcopy_khit_cut", C.c_uint32), ("ocv_w", C.c_uint64)]


class ChainHdr(C.Structure):
    _fields_ = [("n_hits", C.c_uint32), ("w0", C.c_uint32), ("q0", C.c_uint32), ("offset", C.c_uint32), ("pos", C.c_uint64)]


class Delivery(C.Structure):
    """hao_delivery_t: read-only view of one batch's results in a pinned host arena"""
    _fields_ = [("rid_lo", C.c_uint64), ("n_reads", C.c_uint64), ("n_ol", C.c_uint64), ("n_fc", C.c_uint64), ("n_chains", C.c_uint64),
                ("n_cl", C.c_uint64), ("n_exc", C.c_uint64), ("n_codes", C.c_uint64), ("n_pos", C.c_uint64), ("bytes", C.c_uint64),
                ("ol_off", C.c_void_p), ("ol", C.c_void_p), ("fc_off", C.c_void_p), ("fc", C.c_void_p), ("ch_off", C.c_void_p),
                ("cl_off", C.c_void_p), ("qm_off", C.c_void_p), ("chains", C.c_void_p), ("cl_bits", C.c_void_p), ("cl_rank", C.c_void_p), ("cl_codes", C.c_void_p), ("qmz", C.c_void_p),
                ("cl_exc", C.c_void_p), ("exact", C.c_void_p), ("copy_ms", C.c_double), ("qmz_pos", C.c_void_p), ("qmz_cnt", C.c_void_p)]


DELIVER_OL, DELIVER_CL, DELIVER_EXACT = 1, 2, 4

ABI_SYMBOLS = [
    "hao_opt_default", "hao_create", "hao_destroy", "hao_last_error", "hao_set_reads", "hao_ft_gen", "hao_pt_gen",
    "hao_ft_cnt", "hao_pt_get", "hao_ft_table", "hao_pt_table", "hao_hist", "hao_stats", "hao_sketch_batch",
    "hao_fetch_sketch", "hao_overlap_batch", "hao_fetch_seed_hits", "hao_fetch_overlaps", "hao_batch_totals", "hao_batch_seed_path",
    "hao_stage_times", "hao_pass_default", "hao_overlap_batch_ex", "hao_set_shard", "hao_dist_unique_id", "hao_dist_init",
    "hao_loop_create", "hao_loop_destroy", "hao_dist_init_loopback", "hao_batch_digest", "hao_selftest_rocprim", "hao_selftest_big", "hao_selftest_sortbits", "hao_unpack_cigar", "hao_unpack_overlaps", "hao_overlap_batch_async", "hao_deliver_wait", "hao_unpack_hits", "hao_exact_check", "hao_fetch_exact", "hao_window_ed_batch", "hao_index_save", "hao_index_load", "hao_next_slot", "hao_attach", "hao_window_trace_batch", "hao_delivery_digest", "hao_ft_passes", "hao_ovlp_bin_read", "hao_ovlp_bin_write", "hao_window_ed_grid", "hao_fetch_ed_grid",
]


def lib_path():
    return os.path.join(_HERE, "libhao.so")


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise HaoError(f"{p} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(p)
        vp, u8p, u32p, u64p, i64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
        L.hao_opt_default.argtypes = [C.POINTER(Opt)]
        L.hao_create.argtypes = [C.c_int, C.POINTER(Opt), C.POINTER(vp)]
        L.hao_destroy.argtypes = [vp]
        L.hao_attach.argtypes = [vp, C.POINTER(vp)]
        L.hao_last_error.argtypes = [vp]; L.hao_last_error.restype = C.c_char_p
        L.hao_set_reads.argtypes = [vp, u8p, u64p, u32p, C.c_uint64, u64p, u32p]
        L.hao_ft_gen.argtypes = [vp, C.POINTER(C.c_int32)]
        L.hao_pt_gen.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.hao_ft_cnt.argtypes = [vp, C.c_uint64]; L.hao_ft_cnt.restype = C.c_int32
        L.hao_pt_get.argtypes = [vp, C.c_uint64, C.POINTER(u64p), C.POINTER(C.c_int32)]
        L.hao_ft_table.argtypes = [vp, u64p, C.POINTER(u64p), C.POINTER(C.POINTER(C.c_int32))]
        L.hao_pt_table.argtypes = [vp, u64p, C.POINTER(u64p), C.POINTER(u64p), C.POINTER(u64p), u64p]
        L.hao_hist.argtypes = [vp, C.c_int, i64p]
        L.hao_stats.argtypes = [vp, i64p]
        L.hao_ft_passes.argtypes = [vp]; L.hao_ft_passes.restype = C.c_int
        L.hao_sketch_batch.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        L.hao_fetch_sketch.argtypes = [vp, C.c_uint64, C.POINTER(vp), u64p]
        L.hao_overlap_batch.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.hao_pass_default.argtypes = [vp, C.POINTER(Pass)]
        L.hao_overlap_batch_ex.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(Pass)]
        L.hao_fetch_seed_hits.argtypes = [vp, C.c_uint64, C.POINTER(vp), u64p]
        L.hao_fetch_overlaps.argtypes = [vp, C.c_uint64, C.POINTER(vp), u64p, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u64p]
        L.hao_batch_totals.argtypes = [vp, u64p]
        L.hao_batch_seed_path.argtypes = [vp, u64p]
        L.hao_stage_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
        L.hao_batch_digest.argtypes = [vp, u64p, u64p]
        L.hao_overlap_batch_async.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(Pass), C.c_uint32, C.POINTER(C.c_int)]
        L.hao_deliver_wait.argtypes = [vp, C.c_int, C.POINTER(Delivery)]
        L.hao_exact_check.argtypes = [vp]
        L.hao_window_ed_batch.argtypes = [vp, vp, C.c_uint64, vp]
        L.hao_window_trace_batch.argtypes = [vp, C.c_int, vp, C.c_uint64, vp, vp, C.c_uint32]
        L.hao_index_save.argtypes = [vp, C.c_char_p, C.c_int32, vp]
        L.hao_index_load.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int32)]
        L.hao_fetch_exact.argtypes = [vp, C.c_uint64, C.POINTER(vp), u64p]
        L.hao_unpack_hits.argtypes = [C.POINTER(Delivery), C.c_uint64, vp, C.c_uint64]; L.hao_unpack_hits.restype = C.c_uint64
        L.hao_unpack_cigar.argtypes = [C.POINTER(Delivery), C.c_uint64, vp, C.c_uint32]; L.hao_unpack_cigar.restype = C.c_uint32
        L.hao_unpack_overlaps.argtypes = [C.POINTER(Delivery), C.c_uint64, vp, C.c_uint64]; L.hao_unpack_overlaps.restype = C.c_uint64
        L.hao_delivery_digest.argtypes = [C.POINTER(Delivery), u64p, C.c_int]
        L.hao_set_shard.argtypes = [vp, C.c_uint64, C.c_uint64, u32p]
        L.hao_dist_unique_id.argtypes = [u8p]
        L.hao_dist_init.argtypes = [vp, u8p, C.c_int, C.c_int]
        L.hao_loop_create.argtypes = [C.c_int]; L.hao_loop_create.restype = vp
        L.hao_loop_destroy.argtypes = [vp]
        L.hao_dist_init_loopback.argtypes = [vp, vp, C.c_int]
        _LIB = L
    return _LIB


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = int(n) * np.dtype(dtype).itemsize
    addr = ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value
    return np.frombuffer((C.c_uint8 * nbytes).from_address(addr), dtype=dtype).copy()


class Engine:
    """One engine per GPU (one process per GPU).  Options mirror hifiasm_opt_t."""

    def __init__(self, device: int = 0, **opts):
        self.L = lib()
        self.opt = Opt()
        self.L.hao_opt_default(C.byref(self.opt))
        self.bw_thres = opts.pop("bw_thres", None)      # a per-pass argument of h_ec_lchain, not an option: overrides hao_pass_default's value
        for k, v in opts.items():
            if not hasattr(self.opt, k):
                raise HaoError(f"unknown option {k}")
            setattr(self.opt, k, v)
        h = C.c_void_p()
        rc = self.L.hao_create(device, C.byref(self.opt), C.byref(h))
        if rc != 0:
            raise HaoError(f"hao_create failed ({rc}): no usable HIP device - this engine has no CPU fallback")
        self.h = h
        self.n_reads = 0

    def attach(self):
        """hao_attach: a second batch context (own stream, scratch, results) over this engine's reads and index, for a second host thread"""
        v = Engine.__new__(Engine)
        v.L = self.L; v.opt = self.opt; v.bw_thres = self.bw_thres; v.n_reads = self.n_reads; v.owner = self
        h = C.c_void_p()
        self._ck(self.L.hao_attach(self.h, C.byref(h)), "hao_attach")
        v.h = h
        return v

    def close(self):
        if getattr(self, "h", None):
            self.L.hao_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise HaoError(f"{what} failed ({rc}): {self.L.hao_last_error(self.h).decode()}")

    # ---- read store (All_reads, Process_Read.h:115-146) ----
    def set_reads(self, packed, pk_off, lengths, n_mask=None, code_off=None):
        """packed/pk_off/lengths as produced by ha_compress_base; n_mask (per-base 0/1, with code_off) lists N sites."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        pk_off = np.ascontiguousarray(pk_off, dtype=np.uint64)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
        n = lengths.size
        ns_off = ns = None
        if n_mask is not None:
            pos = np.flatnonzero(n_mask).astype(np.uint64)
            co = np.ascontiguousarray(code_off, dtype=np.uint64)
            rid = np.searchsorted(co, pos, side="right") - 1
            ns = (pos - co[rid]).astype(np.uint32)
            ns_off = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(np.bincount(rid, minlength=n), out=ns_off[1:], dtype=np.uint64)
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        self._ck(self.L.hao_set_reads(self.h, packed.ctypes.data_as(u8p), pk_off.ctypes.data_as(u64p), lengths.ctypes.data_as(u32p), n,
                                      ns_off.ctypes.data_as(u64p) if ns_off is not None else None,
                                      ns.ctypes.data_as(u32p) if ns is not None else None), "hao_set_reads")
        self.n_reads = n

    def set_readset(self, rs):
        self.set_reads(rs.packed, rs.pk_off, rs.lengths, rs.n_mask(), rs.code_off)

    # ---- sharded mode (one process per GPU; reads partitioned by query read) ----
    def set_shard(self, rid_base, all_lengths):
        al = np.ascontiguousarray(all_lengths, dtype=np.uint32)
        self._ck(self.L.hao_set_shard(self.h, int(rid_base), al.size, al.ctypes.data_as(C.POINTER(C.c_uint32))), "hao_set_shard")
        self.rid_base = int(rid_base)

    @staticmethod
    def dist_unique_id():
        buf = (C.c_uint8 * 128)()
        if lib().hao_dist_unique_id(buf) != 0:
            raise HaoError("hao_dist_unique_id failed")
        return bytes(buf)

    def dist_init(self, uid: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._ck(self.L.hao_dist_init(self.h, buf, rank, world), "hao_dist_init")

    def dist_init_loopback(self, group, rank: int):
        self._ck(self.L.hao_dist_init_loopback(self.h, group, rank), "hao_dist_init_loopback")

    # ---- ha_ft_gen / ha_pt_gen ----
    def ha_ft_gen(self):
        hom = C.c_int32()
        self._ck(self.L.hao_ft_gen(self.h, C.byref(hom)), "hao_ft_gen")
        return hom.value

    def ha_pt_gen(self):
        hom, het = C.c_int32(), C.c_int32()
        self._ck(self.L.hao_pt_gen(self.h, C.byref(hom), C.byref(het)), "hao_pt_gen")
        return hom.value, het.value

    def ha_ft_cnt(self, y):
        return self.L.hao_ft_cnt(self.h, y)

    def ha_pt_get(self, y):
        p, n = C.POINTER(C.c_uint64)(), C.c_int32()
        self._ck(self.L.hao_pt_get(self.h, y, C.byref(p), C.byref(n)), "hao_pt_get")
        return _arr(p, n.value, np.uint64)

    def ft_table(self):
        n, k, v = C.c_uint64(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_int32)()
        self._ck(self.L.hao_ft_table(self.h, C.byref(n), C.byref(k), C.byref(v)), "hao_ft_table")
        return _arr(k, n.value, np.uint64), _arr(v, n.value, np.int32)

    def pt_table(self):
        nk, npos = C.c_uint64(), C.c_uint64()
        k, o, p = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        self._ck(self.L.hao_pt_table(self.h, C.byref(nk), C.byref(k), C.byref(o), C.byref(p), C.byref(npos)), "hao_pt_table")
        return _arr(k, nk.value, np.uint64), _arr(o, nk.value + 1, np.uint64), _arr(p, npos.value, np.uint64)

    def hist(self, which):
        out = (C.c_int64 * 4096)()
        self._ck(self.L.hao_hist(self.h, which, out), "hao_hist")
        return np.array(out, dtype=np.int64)

    def stats(self):
        out = (C.c_int64 * 8)()
        self._ck(self.L.hao_stats(self.h, out), "hao_stats")
        names = ["ft_peak_hom", "ft_peak_het", "ft_cutoff", "max_n_chain", "hom_cov", "het_cov", "high_occ", "low_occ"]
        return dict(zip(names, [int(x) for x in out]))

    def ft_passes(self):
        """hash-range passes the last ha_ft_gen counted in"""
        return int(self.L.hao_ft_passes(self.h))

    # ---- mz1_ha_sketch ----
    def sketch_batch(self, lo, hi, use_ft=True, sample_dist=None):
        sd = self.opt.sample_dist if sample_dist is None else sample_dist
        self._ck(self.L.hao_sketch_batch(self.h, lo, hi, int(use_ft), sd), "hao_sketch_batch")

    def fetch_sketch(self, rid):
        p, n = C.c_void_p(), C.c_uint64()
        self._ck(self.L.hao_fetch_sketch(self.h, rid, C.byref(p), C.byref(n)), "hao_fetch_sketch")
        return _arr(p.value, 2 * n.value, np.uint64).reshape(-1, 2)

    # ---- h_ec_lchain ----
    def pass_default(self):
        p = Pass()
        self._ck(self.L.hao_pass_default(self.h, C.byref(p)), "hao_pass_default")
        if self.bw_thres is not None:
            p.bw_thres = self.bw_thres
        return p

    def overlap_batch(self, lo, hi, bw_thres=None):
        """h_ec_lchain for reads [lo, hi) with worker_hap_ec's arguments (ecovlp.cpp:3274); bw_thres = 0.001 gives the final-round call (:3957)"""
        if bw_thres is None and self.bw_thres is None:
            self._ck(self.L.hao_overlap_batch(self.h, lo, hi), "hao_overlap_batch")
            return
        p = self.pass_default()
        if bw_thres is not None:
            p.bw_thres = bw_thres
        self._ck(self.L.hao_overlap_batch_ex(self.h, lo, hi, C.byref(p)), "hao_overlap_batch_ex")

    # ---- streaming delivery (hao_overlap_batch_async / hao_deliver_wait / hao_unpack_hits) ----
    def overlap_batch_async(self, lo, hi, parts=DELIVER_OL | DELIVER_CL, bw_thres=None):
        """compute reads [lo, hi) and queue the copy of their results into a pinned host arena; returns the arena slot (0 / 1).  At most two batches are
        in flight: the slot is reused by the second-next async batch."""
        p = None
        if bw_thres is not None or self.bw_thres is not None:
            p = self.pass_default()
            if bw_thres is not None:
                p.bw_thres = bw_thres
        slot = C.c_int(-1)
        self._ck(self.L.hao_overlap_batch_async(self.h, lo, hi, C.byref(p) if p is not None else None, parts, C.byref(slot)), "hao_overlap_batch_async")
        return slot.value

    def deliver_wait(self, slot):
        """the Delivery view of a slot (blocks until its copy has landed)"""
        d = Delivery()
        self._ck(self.L.hao_deliver_wait(self.h, slot, C.byref(d)), "hao_deliver_wait")
        return d

    def delivered_read(self, d, rid):
        """(ol uint32 [n,12], fc uint64, fc_off uint64 [n+1], cl uint32 [m,4]) of read rid out of a Delivery view: what the h_ec_lchain shim hands to its caller"""
        r = rid - d.rid_lo
        oo = _arr(d.ol_off + 8 * r, 2, np.uint64)
        s_, e_ = int(oo[0]), int(oo[1])
        ol = np.zeros((e_ - s_, 12), dtype=np.uint32)                      # (the wire carries 32 of an overlap's 48 bytes)
        got = self.L.hao_unpack_overlaps(C.byref(d), rid, ol.ctypes.data_as(C.c_void_p), e_ - s_)
        assert got == e_ - s_
        lens = ol[:, 11].astype(np.int64)                                  # fc_len of every overlap; the cigars come through the decoder (the wire packs them)
        fo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        fc = np.zeros(int(fo[-1]), dtype=np.uint64)
        for k in range(e_ - s_):
            got = self.L.hao_unpack_cigar(C.byref(d), s_ + k, fc[int(fo[k]):].ctypes.data_as(C.c_void_p), int(lens[k]))
            assert got == int(lens[k])
        co = _arr(d.cl_off + 8 * r, 2, np.uint64)
        m = int(co[1] - co[0])
        cl = np.zeros((m, 4), dtype=np.uint32)
        got = self.L.hao_unpack_hits(C.byref(d), rid, cl.ctypes.data_as(C.c_void_p), m)
        assert got == m
        return ol, fc, fo, cl

    def delivery_digest(self, d, threads=None):
        """hao_batch_digest's per-read value computed on the HOST from a delivered batch: ol, fake cigars, and cl->list decoded out of the wire format"""
        out = np.zeros(int(d.n_reads), dtype=np.uint64)
        self._ck(self.L.hao_delivery_digest(C.byref(d), out.ctypes.data_as(C.POINTER(C.c_uint64)), int(threads or min(32, os.cpu_count() or 1))), "hao_delivery_digest")
        return out

    def index_save(self, prefix, number_of_round=3):
        """write <prefix>.pt_flt / .pt_flt.bin / .pt_flt.paf.bin in the reference's resume format (write_pt_index, htab.cpp:1367)"""
        self._ck(self.L.hao_index_save(self.h, prefix.encode(), number_of_round, None), "hao_index_save")

    def index_load(self, prefix):
        """hao_index_load: read store + filter table + position index from <prefix>.pt_flt[.bin] -> number_of_round stored in the file"""
        r = C.c_int32(0)
        self._ck(self.L.hao_index_load(self.h, prefix.encode(), C.byref(r)), "hao_index_load")
        return r.value

    def window_trace_batch(self, tasks, cap=80, mode=0):
        """tasks: uint32 [n,10] -> (int32 [n,6] (err, ps, pe, ts, te, cigar entries), uint16 [n,cap] cigars): alignment in the band with traceback;
        mode 0 global (ed_band_cal_global_64_w_trace), 1 / 2 forward / backward extension (ed_band_cal_extension_64_{0,1}_w_trace), 3 semi-global with absent
        diagonals (ed_band_cal_semi_64_w_absent_diag_trace)"""
        t = np.ascontiguousarray(tasks, dtype=np.uint32).reshape(-1, 10)
        out = np.zeros((t.shape[0], 6), dtype=np.int32); cig = np.zeros((t.shape[0], cap), dtype=np.uint16)
        self._ck(self.L.hao_window_trace_batch(self.h, mode, t.ctypes.data_as(C.c_void_p), t.shape[0], out.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p), cap),
                 "hao_window_trace_batch")
        return out, cig

    def window_ed_batch(self, tasks):
        """tasks: uint32 [n,10] (p_rid, p_pos, p_len, p_rev, t_rid, t_pos, t_len, t_rev, thre, abs_diag) -> int32 [n,2] (err, pe)"""
        t = np.ascontiguousarray(tasks, dtype=np.uint32).reshape(-1, 10)
        out = np.zeros((t.shape[0], 2), dtype=np.int32)
        self._ck(self.L.hao_window_ed_batch(self.h, t.ctypes.data_as(C.c_void_p), t.shape[0], out.ctypes.data_as(C.c_void_p)), "hao_window_ed_batch")
        return out

    def window_ed_grid(self, window=375, thre=15):
        """window / candidate pairs of the last batch on the reference's window grid, generated and aligned on the device; returns their number"""
        n = C.c_uint64()
        self._ck(self.L.hao_window_ed_grid(self.h, C.c_uint32(window), C.c_uint32(thre), C.byref(n)), "hao_window_ed_grid")
        return int(n.value)

    def fetch_ed_grid(self, n):
        """(tasks uint32 [n,10], results int32 [n,2]) of the last hao_window_ed_grid"""
        t = np.zeros((n, 10), dtype=np.uint32); r = np.zeros((n, 2), dtype=np.int32)
        self._ck(self.L.hao_fetch_ed_grid(self.h, t.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.c_uint64(n)), "hao_fetch_ed_grid")
        return t, r

    def fetch_exact(self, rid):
        """exact-overlap flags (uint8, aligned with h_ec_lchain(rid)[0]) of a read of the last batch"""
        p, n = C.c_void_p(), C.c_uint64()
        self._ck(self.L.hao_fetch_exact(self.h, rid, C.byref(p), C.byref(n)), "hao_fetch_exact")
        return _arr(p.value, n.value, np.uint8)

    def fetch_seed_hits(self, rid):
        p, n = C.c_void_p(), C.c_uint64()
        self._ck(self.L.hao_fetch_seed_hits(self.h, rid, C.byref(p), C.byref(n)), "hao_fetch_seed_hits")
        return _arr(p.value, 4 * n.value, np.uint32).reshape(-1, 4)

    def h_ec_lchain(self, rid):
        """-> (ol uint32 [n,12], fc uint64, fc_off uint64 [n+1], cl uint32 [m,4]) for a read of the last batch."""
        ol, fc, fo, cl = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, m = C.c_uint64(), C.c_uint64()
        self._ck(self.L.hao_fetch_overlaps(self.h, rid, C.byref(ol), C.byref(n), C.byref(fc), C.byref(fo), C.byref(cl), C.byref(m)),
                 "hao_fetch_overlaps")
        foff = _arr(fo.value, n.value + 1, np.uint64)
        nfc = int(foff[-1] - foff[0]) if foff.size else 0
        return (_arr(ol.value, 12 * n.value, np.uint32).reshape(-1, 12), _arr(fc.value, nfc, np.uint64), foff - (foff[0] if foff.size else 0),
                _arr(cl.value, 4 * m.value, np.uint32).reshape(-1, 4))

    def batch_totals(self):
        out = (C.c_uint64 * 8)()
        self._ck(self.L.hao_batch_totals(self.h, out), "hao_batch_totals")
        return dict(overlaps=int(out[0]), chained_hits=int(out[1]), seed_hits=int(out[2]), groups=int(out[3]), minimizers=int(out[4]), chains=int(out[5]),
                    seq_groups=int(out[6]), seq_group_hits=int(out[7]))

    def batch_seed_path(self):
        """which kernels carried the last batch's seed stage: (first launch: 2 list-major / 1 one-wave merge / 0 table kernels, reads left to the tables, 512- / 1024-slot overflows)"""
        out = (C.c_uint64 * 4)()
        self._ck(self.L.hao_batch_seed_path(self.h, out), "hao_batch_seed_path")
        return dict(first_launch={0: "seed_bin_kernel", 1: "(unused)", 2: "seed_lds_kernel"}[int(out[0])], left_to_tables=int(out[1]), overflow_512=int(out[2]), overflow_1024=int(out[3]))

    def batch_digest(self, n, with_seed_hits=True):
        """per-read digests of the last batch (n reads): (digest of ol / fake cigars / cl, digest of the seed hits or None)"""
        d = np.zeros(n, dtype=np.uint64)
        k = np.zeros(n, dtype=np.uint64) if with_seed_hits else None
        u64p = C.POINTER(C.c_uint64)
        self._ck(self.L.hao_batch_digest(self.h, d.ctypes.data_as(u64p), k.ctypes.data_as(u64p) if k is not None else None), "hao_batch_digest")
        return d, k

    def stage_times(self):
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        n = self.L.hao_stage_times(self.h, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]
