"""Every FALLBACK path of libhao.so that a run-time switch can force on every read / group (HAO_DBG_FORCE, HAO_DBG_TEST, the seed stage's
HAO_SEED_* thresholds: hao_ctx.hpp) must give the oracle's result: one-lane sequential chaining instead of the wave kernels, DP without
speculative tiles, the one-lane DP tail, DP and pack kernels on the engine's stream, the one-lane pruning scan, the sketch retry after an
under-sized minimizer pool, the seed stage's generic tables, and the seed
stage's table kernels / list-major kernel on every batch."""
import os

import pytest

from helpers import scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu

SWITCHES = ["HAO_DBG_FORCE=seq_chain", "HAO_DBG_FORCE=dp_nospec", "HAO_DBG_FORCE=dp_seqtail", "HAO_DBG_FORCE=dp_serial", "HAO_DBG_FORCE=seq_prune", "HAO_DBG_FORCE=noql",
            "HAO_DBG_TEST=sk_gcap=1000",
            "HAO_SEED_LDS=0", "HAO_SEED_LDS_RATIO=1000000", "HAO_SEED_LDS_RATIO=1", "HAO_SEED_MERGE_MAXN=3000", "HAO_SEED_MERGE_MAXN=1000000"]
# (the big-index build on small sets - HAO_DBG_TEST=sort40_min=1 - runs on the CPU emulation only, tests/test_simt_schedules_cpu.py: rocprim's bit-range sort
# mis-sorts inputs of 5 k - 200 k elements on this ROCm, tests/test_gpu_rocprim.py.  The seed stage: the table kernels for every read; the list-major kernel for every batch however many hits its reads average - these sets are repeat-rich - and the seed-hit limit
# above which it leaves a read to the table kernels)


def _env_of(switch):
    """'A=x' -> {A: x} (x may hold '=' and ',': the lists of HAO_DBG_FORCE / HAO_DBG_TEST); several variables: 'A=x;B=y'"""
    return dict(kv.split("=", 1) for kv in switch.split(";"))


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
    """List starts are 48-bit everywhere (lk[], the key table, the seed kernels' staged words, the host view): HAO_DBG_TEST=ix_pad=N puts 2^32 + 12345 unused position
    records in front of the index (34 GB), so every list of a small read set starts beyond 2^32 - the situation of a replicated index of more than 2^32
    minimizers (human genome, 50x).  Same tables, same overlaps."""
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    os.environ["HAO_DBG_TEST"] = "ix_pad=" + str((1 << 32) + 12345)
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
        del os.environ["HAO_DBG_TEST"]
    assert bad == 0, f"{bad}/{rs.n} reads differ"
