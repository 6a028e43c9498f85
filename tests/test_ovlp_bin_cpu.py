"""SURVEY 8 f4, second half: the reference's *.ovlp.source.bin / *.ovlp.reverse.bin (write_ma_hit_ts / load_ma_hit_ts, Overlaps.cpp:23328-23469) through
hao_ovlp_bin_read / hao_ovlp_bin_write.  The UNMODIFIED reference executable writes the files for a small read set; the reader must take them apart (one list of 42-byte
records per read), the writer must put them together again byte for byte, and damaged files must be refused.  Host code only: no device is involved."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "hifiasm_ref")


class MaHit(C.Structure):      # hao_ma_hit_t (include/hao.h)
    _fields_ = [("qns", C.c_uint64), ("qe", C.c_uint32), ("tn", C.c_uint32), ("ts", C.c_uint32), ("te", C.c_uint32), ("ml", C.c_uint32), ("rev", C.c_uint32),
                ("bl", C.c_uint32), ("del_", C.c_uint32), ("el", C.c_uint8), ("no_l_indel", C.c_uint8), ("pad", C.c_uint8 * 6)]


def _read(L, path):
    n = C.c_uint64(); fl = C.POINTER(C.c_uint8)(); off = C.POINTER(C.c_uint64)(); hits = C.POINTER(MaHit)()
    rc = L.hao_ovlp_bin_read(path.encode(), C.byref(n), C.byref(fl), C.byref(off), C.byref(hits))
    return rc, n.value, fl, off, hits


@pytest.fixture(scope="module")
def bins(tmp_path_factory):
    if not os.path.exists(REF):
        pytest.skip("the reference executable is built only where /root/reference exists")
    from hifiasm_amd import synth
    d = str(tmp_path_factory.mktemp("ovlp_bin"))
    rs = synth.dataset(genome_size=120_000, coverage=25, read_len=8000, err=0.002, seed=9, len_jit=2000)
    fa = os.path.join(d, "reads.fa"); synth.write_fasta(fa, rs)
    r = subprocess.run([REF, "-o", os.path.join(d, "ref"), "-t", "4", "-f0", "--bin-only", fa], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-800:]
    return d, rs.n


def test_read_and_write_back(bins):
    from hifiasm_amd import api
    d, n_reads = bins
    L = api.lib()
    L.hao_ovlp_bin_read.restype = C.c_int; L.hao_ovlp_bin_write.restype = C.c_int
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    assert C.sizeof(MaHit) == 48
    for ext in ("ovlp.source.bin", "ovlp.reverse.bin"):
        src = os.path.join(d, "ref." + ext)
        rc, n, fl, off, hits = _read(L, src)
        assert rc == 0 and n == n_reads
        o = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
        assert o[0] == 0 and (np.diff(o.astype(np.int64)) >= 0).all()
        raw = open(src, "rb").read()
        assert len(raw) == 8 + 6 * n + 42 * int(o[n])
        if ext == "ovlp.source.bin":
            assert o[n] > 1000
            # a record of read i names read i as its query (qns = query id << 32 | query start) and another read as its target
            i = int(np.argmax(np.diff(o.astype(np.int64)))); h = hits[int(o[i])]
            assert (h.qns >> 32) == i and h.tn != i and h.tn < n and h.ts <= h.te and h.rev in (0, 1)
        out = os.path.join(d, "hao." + ext)
        assert L.hao_ovlp_bin_write(out.encode(), C.c_uint64(n), fl, off, hits) == 0
        assert open(out, "rb").read() == raw, ext
        for p in (fl, off, hits):
            libc.free(C.cast(p, C.c_void_p))


def test_damaged_files_are_refused(bins, tmp_path):
    from hifiasm_amd import api
    d, _ = bins
    L = api.lib(); L.hao_ovlp_bin_read.restype = C.c_int
    raw = open(os.path.join(d, "ref.ovlp.source.bin"), "rb").read()
    cases = {"truncated": raw[:len(raw) // 2], "trailing": raw + b"\x00", "count": (10 ** 15).to_bytes(8, "little") + raw[8:], "negative": (-5).to_bytes(8, "little", signed=True) + raw[8:],
             "length": raw[:10] + (0x7fffffff).to_bytes(4, "little") + raw[14:], "empty": b""}
    for name, data in cases.items():
        p = str(tmp_path / (name + ".bin")); open(p, "wb").write(data)
        rc, n, fl, off, hits = _read(L, p)
        assert rc != 0 and n == 0 and not fl and not off and not hits, name
    assert _read(L, str(tmp_path / "missing.bin"))[0] != 0
