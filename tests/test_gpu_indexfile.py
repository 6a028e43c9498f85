"""On-disk formats: the engine's tables and read store written by hao_index_save in the reference's resume format are loaded by the
UNMODIFIED reference (load_pt_index, htab.cpp:1432, through oracle/_ref/ref_harness --load-index); everything the reference then computes from the
loaded index - filter table, position index, minimizers, overlap lists, fake cigars, chained hits - must equal the golden dump it produced when it
built the index itself from the FASTA."""
import os
import subprocess
import tempfile
import zlib

import numpy as np
import pytest

import oracle_py
from helpers import scenario_reads, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


@pytest.mark.parametrize("name", ["hifi", "rr", "nn", "k40", "ont", "edge"])
def test_reference_loads_the_gpu_built_index(name):
    if not os.path.exists(HARNESS):
        pytest.skip("oracle/_ref/ref_harness not built (needs /root/reference at build time)")
    from hifiasm_amd import synth
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    g = load_golden(name)
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    d = tempfile.mkdtemp(prefix="hao_idx_")
    e.index_save(os.path.join(d, "gpu"))
    e.close()
    for suffix in (".pt_flt", ".pt_flt.bin", ".pt_flt.paf.bin"):
        assert os.path.getsize(os.path.join(d, "gpu" + suffix)) > 0
    ont = bool(okw.get("is_ont"))
    fa = os.path.join(d, "r.fq" if ont else "r.fa")
    synth.write_fasta(fa, rs, fastq=ont)                      # (only to satisfy the option parser: the loader never opens it)
    cmd = [HARNESS, "-t", "2", "--load-index", os.path.join(d, "gpu"), "--dump", os.path.join(d, "s")]
    cmd += ["--ont"] if ont else []
    for k_, flag in (("k", "-k"), ("w", "-w")):
        if k_ in okw:
            cmd += [flag, str(okw[k_])]
    r = subprocess.run(cmd + [fa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    dump = oracle_py.load_ref_dump(os.path.join(d, "s"))
    for key in ("ft_keys", "ft_vals", "pt_keys", "pt_off", "pt_pos", "mz_off", "mz", "ol_off", "ol", "fc_off", "fc", "kh_off", "cl_off", "rlen", "ex"):
        assert dump[key].shape == g[key].shape and (dump[key] == g[key]).all(), key
    cl = dump["cl"].reshape(-1, 4)
    crc = np.array([zlib.crc32(cl[int(dump["cl_off"][i]):int(dump["cl_off"][i + 1])].tobytes()) for i in range(rs.n)], dtype=np.uint64)
    assert (crc == g["cl_crc"]).all()
    for key in ("hom_cov", "het_cov", "max_n_chain", "high_occ", "low_occ"):
        assert dump["meta"][key] == g["meta"][key], key


@pytest.mark.parametrize("name", ["hifi", "rr", "nn", "ont", "edge"])
def test_engine_loads_the_reference_built_index(name):
    """the other direction: the UNMODIFIED reference builds the index from the FASTA and writes it (write_pt_index, htab.cpp:1367, through
    ref_harness --save-index); hao_index_load makes it the engine's state - read store, filter table, position index, peaks - and everything the
    engine then computes equals the golden dump / the oracle: tables, thresholds, minimizers, every read's overlaps, fake cigars and chained hits"""
    if not os.path.exists(HARNESS):
        pytest.skip("oracle/_ref/ref_harness not built (needs /root/reference at build time)")
    from hifiasm_amd import synth
    from hifiasm_amd.api import Engine
    from helpers import scenario_oracle
    rs, okw = scenario_reads(name)
    g = load_golden(name)
    d = tempfile.mkdtemp(prefix="hao_idx_")
    ont = bool(okw.get("is_ont"))
    fa = os.path.join(d, "r.fq" if ont else "r.fa")
    synth.write_fasta(fa, rs, fastq=ont)
    cmd = [HARNESS, "-t", "2", "--save-index", os.path.join(d, "ref")] + (["--ont"] if ont else [])
    r = subprocess.run(cmd + [fa], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    e = Engine(0, **okw)
    assert e.index_load(os.path.join(d, "ref")) == 3                       # the reference's default -r
    keys, vals = e.ft_table()
    assert keys.shape == g["ft_keys"].shape and (keys == g["ft_keys"]).all() and (vals == g["ft_vals"]).all()
    pk, po, pp = e.pt_table()
    assert (pk == g["pt_keys"]).all() and (po == g["pt_off"]).all() and (pp == g["pt_pos"]).all()
    st = e.stats()
    for key in ("hom_cov", "het_cov", "max_n_chain", "high_occ", "low_occ"):
        assert st[key] == g["meta"][key], key
    o = scenario_oracle(name)
    e.overlap_batch(0, rs.n)
    for rid in range(rs.n):
        ol, fc, fo, cl = e.h_ec_lchain(rid)
        ool, ofc, ofo, ocl = o.lchain(rid)
        assert ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and cl.shape == ocl.shape and (cl == ocl).all(), rid
    e.sketch_batch(0, rs.n)
    for rid in range(0, rs.n, 7):
        assert (e.fetch_sketch(rid) == o.sketch(rid)).all(), rid
    e.close()


def test_damaged_index_files_are_refused():
    """hao_index_load sizes its buffers from fields of the file: a truncated or overwritten file must come back as an error code (never an exception through the C
    boundary or a crash), leave the engine without an index, and the same engine must still load the intact files afterwards."""
    from hifiasm_amd.api import Engine, HaoError
    rs, okw = scenario_reads("hifi")
    e = Engine(0, **okw)
    e.set_readset(rs)
    e.ha_ft_gen(); e.ha_pt_gen()
    d = tempfile.mkdtemp(prefix="hao_idx_")
    good = os.path.join(d, "good")
    e.index_save(good)
    e.overlap_batch(0, rs.n)
    want = [e.h_ec_lchain(r)[0].copy() for r in range(0, rs.n, 7)]
    e.close()
    blob = {s: open(good + s, "rb").read() for s in (".pt_flt", ".pt_flt.bin", ".pt_flt.paf.bin")}

    def variant(tag, which, data):
        p = os.path.join(d, tag)
        for s, b in blob.items():
            open(p + s, "wb").write(data if s == which else b)
        return p

    pt, rb = blob[".pt_flt"], blob[".pt_flt.bin"]
    huge = (1 << 31).to_bytes(4, "little")
    cases = [variant("cut_pt", ".pt_flt", pt[: len(pt) // 2]), variant("cut_pt2", ".pt_flt", pt[:40]), variant("cut_reads", ".pt_flt.bin", rb[: len(rb) // 3]),
             variant("buckets", ".pt_flt", pt[:1] + huge + pt[5:]),                      # n_buckets of the filter table = 2^31
             variant("count", ".pt_flt", pt[:9] + huge + pt[13:]),                       # more keys than buckets
             variant("nreads", ".pt_flt.bin", rb[:20] + (1 << 27).to_bytes(8, "little") + rb[28:])]      # 2^27 reads in a file of a few hundred
    e = Engine(0, **okw)
    for p in cases:
        with pytest.raises(HaoError):
            e.index_load(p)
        with pytest.raises(HaoError):                        # no index after a failed load
            e.overlap_batch(0, 1)
    e.index_load(good)
    e.overlap_batch(0, rs.n)
    got = [e.h_ec_lchain(r)[0] for r in range(0, rs.n, 7)]
    e.close()
    assert all(a.shape == b.shape and (a == b).all() for a, b in zip(got, want))
