"""The UNMODIFIED reference executable served by the emulated device library: oracle/_ref/hifiasm_hao (the reference's objects + integration/hao_hifiasm_shim.cpp,
linked against libhao.so) runs with tests/simt/_build/libhao_simt.so pre-loaded in libhao.so's place - `ha_ft_gen`, `ha_pt_gen` and every `h_ec_lchain` of its
three correction rounds and its final overlap round, under 8 worker threads, through the streaming delivery - and must write the same *.ovlp.source.bin /
*.ovlp.reverse.bin / *.ec.bin as the reference on its own.  The CPU twin of tests/test_gpu_dropin.py on a smaller read set."""
import os
import subprocess
import tempfile

import pytest

import simt_build
import test_gpu_dropin as T
from simt_suite import FULL


@pytest.mark.skipif(not (os.path.exists(T.REF) and os.path.exists(T.HAO)), reason="reference binaries not built (needs /root/reference at build time)")
@pytest.mark.parametrize("bf,shim_batch", [("-f0", "64")] + ([("-f26", "257"), ("-f0", "4096")] if FULL else []))
def test_bins_identical_on_the_emulated_library(bf, shim_batch):
    from hifiasm_amd import synth
    lib = simt_build.build_lib()
    rs = synth.dataset(genome_size=40_000, coverage=18, read_len=4000, err=0.001, seed=42, len_jit=1000)
    d = tempfile.mkdtemp(prefix="hao_dropin_simt_")
    fa = os.path.join(d, "reads.fa")
    synth.write_fasta(fa, rs)
    for exe, tag in ((T.REF, "ref"), (T.HAO, "hao")):
        env = dict(os.environ, HAO_SHIM_BATCH=shim_batch, **({"HAO_SHIM_FINAL_OL_ONLY": "1"} if bf == "-f26" else {}))
        if tag == "hao":
            env["LD_PRELOAD"] = lib      # its symbols come first: the executable never reaches libhao.so (there is no GPU here)
        r = subprocess.run([exe, "-o", os.path.join(d, tag), "-t", "8", bf, "--bin-only", fa], capture_output=True, text=True, cwd=d, env=env)
        assert r.returncode == 0, f"{tag} failed: {r.stderr[-1500:]}"
    for ext in ("ovlp.source.bin", "ovlp.reverse.bin"):
        a = open(os.path.join(d, f"ref.{ext}"), "rb").read()
        b = open(os.path.join(d, f"hao.{ext}"), "rb").read()
        assert len(a) > 1000 and a == b, f"{ext} differs ({len(a)} vs {len(b)} bytes)"
    a = bytearray(open(os.path.join(d, "ref.ec.bin"), "rb").read())
    b = bytearray(open(os.path.join(d, "hao.ec.bin"), "rb").read())
    assert len(a) == len(b)
    for i in T._ec_mask(bytes(a)):
        a[i] = b[i] = 0
    assert a == b, "ec.bin differs outside the reference's uninitialised pad bytes"
