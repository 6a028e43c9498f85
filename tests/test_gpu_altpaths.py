"""Every implementation variant that libhao.so can be switched to at run time (A/B switches kept for measurements, and the fallbacks behind
them) must give the oracle's result: one-lane sequential chaining instead of the wave kernels, DP without speculative tiles, the one-lane DP
tail, DP kernels on the main stream, one-wave selection for every size, the one-lane pruning scan, the generic (any w, k) sketch kernel,
the sketch retry after an under-sized minimizer list, and the seed stage's table kernels / list-major kernel on every batch."""
import os

import pytest

from helpers import scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu

SWITCHES = ["HAO_DBG_SEQ_CHAIN", "HAO_DBG_DP_NOSPEC", "HAO_DBG_DP_SEQTAIL", "HAO_DBG_DP_SERIAL", "HAO_DBG_SEL1", "HAO_DBG_SEQ_PRUNE",
            "HAO_DBG_SK_GENERIC", "HAO_DBG_SK_GCAP", "HAO_SPEC_MINCLS", "HAO_CHAIN_WPB", "HAO_DBG_TINY_LANE",
            "HAO_SEED_NOQL", "HAO_PT_SORT64", "HAO_PT_DIRECT",
            "HAO_SEED_LDS=0", "HAO_SEED_LDS_RATIO=1000000", "HAO_SEED_LDS_RATIO=1", "HAO_SEED_MERGE_MAXN=3000", "HAO_SEED_MERGE_MAXN=1000000"]
# (the seed stage: the table kernels for every read; the list-major kernel for every batch however many hits its reads average - these sets are repeat-rich - and the
# seed-hit limit above which it leaves a read to the table kernels)
VALUES = {"HAO_DBG_SK_GCAP": "1000", "HAO_SPEC_MINCLS": "0", "HAO_CHAIN_WPB": "4"}


def _env_of(switch):
    """'A' -> {A: VALUES.get(A, '1')}; 'A=x,B=y' -> {A: x, B: y}"""
    if "=" not in switch:
        return {switch: VALUES.get(switch, "1")}
    return dict(kv.split("=") for kv in switch.split(","))


@pytest.mark.parametrize("switch", SWITCHES)
@pytest.mark.parametrize("name", ["rr", "rr_heavy"])
def test_switch_keeps_results(name, switch):
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    env = _env_of(switch)
    os.environ.update(env)
    try:
        e = Engine(0, **okw)
        e.set_readset(rs)
        e.ha_ft_gen()
        e.ha_pt_gen()
        e.overlap_batch(0, rs.n)
        bad = 0
        for r in range(rs.n):
            ol, fc, fo, cl = e.h_ec_lchain(r)
            ool, ofc, ofo, ocl = o.lchain(r)
            if not (ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and (fo == ofo).all() and cl.shape == ocl.shape and (cl == ocl).all()):
                bad += 1
        e.close()
    finally:
        for k in env:
            del os.environ[k]
    assert bad == 0, f"{switch}: {bad}/{rs.n} reads differ"


@pytest.mark.parametrize("name", ["hifi", "rr"])
def test_index_positions_beyond_2_32(name):
    """List starts are 48-bit everywhere (lk[], the key table, the seed kernels' staged words, the host view): HAO_DBG_IX_PAD puts 2^32 + 12345 unused position
    records in front of the index (34 GB), so every list of a small read set starts beyond 2^32 - the situation of a replicated index of more than 2^32
    minimizers (human genome, 50x).  Same tables, same overlaps."""
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    os.environ["HAO_DBG_IX_PAD"] = str((1 << 32) + 12345)
    try:
        e = Engine(0, **okw)
        e.set_readset(rs)
        e.ha_ft_gen()
        e.ha_pt_gen()
        k, off, pos = e.pt_table()
        ok, ooff, opos = o.pt_table()
        assert k.shape == ok.shape and (k == ok).all() and (off == ooff).all() and (pos == opos).all()
        i = ok.size // 2
        assert (e.ha_pt_get(int(ok[i])) == opos[int(ooff[i]):int(ooff[i + 1])]).all()
        e.overlap_batch(0, rs.n)
        bad = 0
        for r in range(rs.n):
            ol, fc, fo, cl = e.h_ec_lchain(r)
            ool, ofc, ofo, ocl = o.lchain(r)
            if not (ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and (fo == ofo).all() and cl.shape == ocl.shape and (cl == ocl).all()):
                bad += 1
        e.close()
    finally:
        del os.environ["HAO_DBG_IX_PAD"]
    assert bad == 0, f"{bad}/{rs.n} reads differ"
