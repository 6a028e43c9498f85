"""Seeded synthetic HiFi / ONT read sets (BASELINE.json configs 2-5).

Thin ctypes wrapper over csrc/hao_synth.c (bench + test tooling).  Base codes are
0..3 = A,C,G,T (4 = N); the packed form is the reference read-store layout
(4 bases/byte, first base in the two most significant bits, len/4+1 bytes per
read, N stored as A - ha_compress_base, Process_Read.cpp:792-850).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libhaosynth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built - run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.hao_synth_genome.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_int]
        L.hao_synth_genome.restype = None
        L.hao_synth_read_lengths.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, C.c_uint64, u32p]
        L.hao_synth_read_lengths.restype = None
        L.hao_synth_reads.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint64, u8p, u64p, u8p, u64p]
        L.hao_synth_reads.restype = None
        L.hao_synth_fasta.argtypes = [u8p, u64p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_int]
        L.hao_synth_fasta.restype = C.c_uint64
        L.hao_synth_fasta_file.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_uint64, C.c_char_p, C.c_int, C.c_int]
        L.hao_synth_fasta_file.restype = C.c_uint64
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def make_genome(size: int, seed: int = 11, repeat_rich: int = 0) -> np.ndarray:
    g = np.empty(size, dtype=np.uint8)
    _lib().hao_synth_genome(_p(g, C.c_uint8), size, seed, int(repeat_rich))
    return g


class ReadSet:
    """A batch of reads [rid0, rid0+n): packed 2-bit store + optional codes."""

    def __init__(self, rid0, lengths, packed, pk_off, codes=None, code_off=None):
        self.rid0 = rid0
        self.lengths = lengths      # uint32 [n]
        self.packed = packed        # uint8, reference read-store layout
        self.pk_off = pk_off        # uint64 [n+1] byte offsets into packed
        self.codes = codes          # uint8 0..4 or None
        self.code_off = code_off    # uint64 [n+1]

    @property
    def n(self):
        return int(self.lengths.size)

    @property
    def total_bases(self):
        return int(self.lengths.sum(dtype=np.uint64))

    def n_mask(self):
        """per-base 0/1 mask of N positions, or None when the set is N-free."""
        if self.codes is None:
            return None
        m = self.codes > 3
        return m.astype(np.uint8) if m.any() else None


def make_reads(genome: np.ndarray, n_reads: int, read_len: int, err: float, seed: int = 7, rid0: int = 0,
               len_jit: int = 0, n_rate: float = 0.0, want_codes: bool = True) -> ReadSet:
    L = _lib()
    err_ppm = int(round(err * 1e6))
    n_ppm = int(round(n_rate * 1e6))
    lens = np.empty(n_reads, dtype=np.uint32)
    L.hao_synth_read_lengths(_p(genome, C.c_uint8), genome.size, rid0, n_reads, read_len, len_jit, err_ppm, n_ppm, seed,
                             _p(lens, C.c_uint32))
    code_off = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=code_off[1:], dtype=np.uint64)
    pk_off = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens // 4 + 1, out=pk_off[1:], dtype=np.uint64)
    codes = np.empty(int(code_off[-1]), dtype=np.uint8) if want_codes else None
    packed = np.empty(int(pk_off[-1]), dtype=np.uint8)
    L.hao_synth_reads(_p(genome, C.c_uint8), genome.size, rid0, n_reads, read_len, len_jit, err_ppm, n_ppm, seed,
                      _p(codes, C.c_uint8), _p(code_off, C.c_uint64), _p(packed, C.c_uint8), _p(pk_off, C.c_uint64))
    return ReadSet(rid0, lens, packed, pk_off, codes, code_off if want_codes else None)


def from_codes(reads, rid0: int = 0) -> ReadSet:
    """ReadSet from explicit base-code arrays (0..3, >= 4 = N): the reference read-store packing (4 bases per byte, first base in
    bits 7..6, N stored as A; ha_compress_base, Process_Read.cpp:792-850).  For hand-made edge-case sets."""
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    code_off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens, out=code_off[1:], dtype=np.uint64)
    pk_off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens // 4 + 1, out=pk_off[1:], dtype=np.uint64)
    codes = np.concatenate([np.asarray(r, dtype=np.uint8) for r in reads]) if reads else np.zeros(0, dtype=np.uint8)
    packed = np.zeros(int(pk_off[-1]), dtype=np.uint8)
    for i, r in enumerate(reads):
        c = np.asarray(r, dtype=np.uint8).copy()
        c[c > 3] = 0
        pad = (-len(c)) % 4
        q = np.concatenate([c, np.zeros(pad, dtype=np.uint8)]).reshape(-1, 4)
        b = (q[:, 0] << 6 | q[:, 1] << 4 | q[:, 2] << 2 | q[:, 3]).astype(np.uint8)
        packed[int(pk_off[i]):int(pk_off[i]) + b.size] = b
    return ReadSet(rid0, lens, packed, pk_off, codes, code_off)


def dataset(genome_size: int, coverage: float, read_len: int, err: float, seed: int = 11, repeat_rich: int = 0,
            len_jit: int = 0, n_rate: float = 0.0, want_codes: bool = True) -> ReadSet:
    g = make_genome(genome_size, seed=seed, repeat_rich=repeat_rich)
    n_reads = max(1, int(round(genome_size * coverage / read_len)))
    return make_reads(g, n_reads, read_len, err, seed=seed + 1, len_jit=len_jit, n_rate=n_rate, want_codes=want_codes)


def write_fasta(path: str, rs: ReadSet, fastq: bool = False, qual: int = 20) -> None:
    assert rs.codes is not None
    n = rs.n
    cap = int(rs.code_off[-1]) * (2 if fastq else 1) + n * 40 + 16
    buf = C.create_string_buffer(cap)
    w = _lib().hao_synth_fasta(_p(rs.codes, C.c_uint8), _p(rs.code_off, C.c_uint64), rs.rid0, n, buf, int(fastq), qual)
    with open(path, "wb") as fp:
        fp.write(buf.raw[:w])


def write_fasta_stream(path: str, genome: np.ndarray, n_reads: int, read_len: int, err: float, seed: int = 7, rid0: int = 0,
                       len_jit: int = 0, n_rate: float = 0.0, fastq: bool = False, qual: int = 20) -> int:
    """FASTA/FASTQ of the reads make_reads() would return, written read by read (no codes array in memory)."""
    w = _lib().hao_synth_fasta_file(_p(genome, C.c_uint8), genome.size, rid0, n_reads, read_len, len_jit, int(round(err * 1e6)),
                                    int(round(n_rate * 1e6)), seed, path.encode(), int(fastq), qual)
    if w == 0:
        raise OSError(f"cannot write {path}")
    return int(w)
