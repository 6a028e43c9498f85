"""End-to-end drop-in check (BASELINE.json configs[0], "plumbing"): the UNMODIFIED reference executable
vs the same objects with ha_ft_gen / ha_pt_gen / h_ec_lchain (+ accessors) served by libhao.so through
integration/hao_hifiasm_shim.cpp.  Both run `--bin-only` at `-f0` and with a Bloom pre-filter (`-f26`); the three bins must agree:
*.ovlp.source.bin and *.ovlp.reverse.bin byte for byte, *.ec.bin except the reference's own uninitialised
bytes: the pad byte at read_sperate[i][len/4] when len % 4 == 0 and the never-written
tail of name_index[] beyond total_reads+1 entries.

Needs the binaries built in the build container (oracle/Makefile target hao-hifiasm); they travel with the
snapshot.  Skipped when absent."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "hifiasm_ref")
HAO = os.path.join(ROOT, "oracle", "_ref", "hifiasm_hao")


def _ec_mask(buf):
    """indices of the pad bytes in a *.ec.bin image (write_All_reads, Process_Read.cpp:69-125)"""
    o = 4
    index_size, name_index_size, total_reads, total_bases, total_name = struct.unpack_from("<5Q", buf, o)
    o += 40
    for _ in range(total_reads):
        (nn,) = struct.unpack_from("<Q", buf, o)
        o += 8 + 8 * nn
    lens = np.frombuffer(buf, dtype="<u8", count=total_reads, offset=o)
    o += 8 * total_reads
    mask = []
    for L in lens:
        L = int(L)
        if L % 4 == 0:
            mask.append(o + L // 4)
        o += L // 4 + 1
    # names blob, then name_index[name_index_size]: only entries 0..total_reads are ever written
    # (ha_insert_read_len, Process_Read.cpp:414-430); the realloc'ed tail is uninitialised heap in the reference
    o += total_name
    for e in range(total_reads + 1, name_index_size):
        mask.extend(range(o + 8 * e, o + 8 * e + 8))
    return mask


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HAO)), reason="reference binaries not built")
@pytest.mark.parametrize("bf,shim_batch", [("-f0", "4096"), ("-f26", "257"), ("-f37", "64")])      # exact counting / a small Bloom pre-filter / the reference's DEFAULT 2^37-bit filter;
def test_bins_identical(bf, shim_batch):                                                               # shim batches of 4096 / 257 / 64 reads: one batch ... dozens of batches streamed under 8 worker threads
    from hifiasm_amd import synth
    rs = synth.dataset(genome_size=300_000, coverage=30, read_len=12000, err=0.001, seed=42, len_jit=3000)
    d = tempfile.mkdtemp(prefix="hao_dropin_")
    fa = os.path.join(d, "reads.fa")
    synth.write_fasta(fa, rs)
    for exe, tag in ((REF, "ref"), (HAO, "hao")):
        r = subprocess.run([exe, "-o", os.path.join(d, tag), "-t", "8", bf, "--bin-only", fa], capture_output=True, text=True, cwd=d,
                           env=dict(os.environ, HAO_SHIM_BATCH=shim_batch, **({"HAO_SHIM_FINAL_OL_ONLY": "1"} if bf == "-f26" else {})))      # (-f26: the final round served without its chained hits, the shim's opt-in)
        assert r.returncode == 0, f"{tag} failed: {r.stderr[-1500:]}"
    for ext in ("ovlp.source.bin", "ovlp.reverse.bin"):
        a = open(os.path.join(d, f"ref.{ext}"), "rb").read()
        b = open(os.path.join(d, f"hao.{ext}"), "rb").read()
        assert len(a) > 1000 and a == b, f"{ext} differs ({len(a)} vs {len(b)} bytes)"
    a = bytearray(open(os.path.join(d, "ref.ec.bin"), "rb").read())
    b = bytearray(open(os.path.join(d, "hao.ec.bin"), "rb").read())
    assert len(a) == len(b)
    for i in _ec_mask(bytes(a)):
        a[i] = b[i] = 0
    assert a == b, "ec.bin differs outside the reference's uninitialised pad bytes"
