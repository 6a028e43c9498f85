"""The algorithm of the unit sketch kernel (tests/sk3_model.py = hao_sketch3.cuh restated lane by lane) against the oracle's mz1_ha_sketch
restatement (pinned to the reference by test_oracle_golden.py): proxies of 32, 12 and 4 bits (the coarse ones flood the verification
with false candidates), with and without the count order of a filter table, reads of every length class of the `edge` scenario."""
import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle
import sk3_model as M


def _cmp(o, rs, rid, use_ft, bits):
    codes = rs.codes[int(rs.code_off[rid]):int(rs.code_off[rid + 1])]
    if (codes > 3).any():
        return None
    ref = o.sketch(rid, use_ft=use_ft, sample_dist=0)
    ft = (lambda y: int(o.L.hao_or_ft_cnt(o.h, y))) if use_ft else None
    got = M.sketch_read(codes, True, ft, bits)
    info = ref[:, 1]
    exp = np.stack([ref[:, 0], (info >> np.uint64(28)) & np.uint64(0x7ffffff), ((info >> np.uint64(55)) & np.uint64(1)) | ((info >> np.uint64(56)) << np.uint64(8))], axis=1)
    return got[:, :3], exp


@pytest.mark.parametrize("name,use_ft,bits,step", [("hifi", False, 32, 9), ("hifi", False, 4, 17), ("edge", False, 32, 1), ("edge", False, 8, 3),
                                                     ("rr", True, 32, 15), ("rr", True, 12, 23), ("rr", True, 6, 31)])
def test_model_equals_oracle(name, use_ft, bits, step):
    import ctypes as C
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    o.L.hao_or_ft_cnt.restype = C.c_int32; o.L.hao_or_ft_cnt.argtypes = [C.c_void_p, C.c_uint64]
    rids = list(range(0, rs.n, step))
    if use_ft:                                  # reads that really contain high-count k-mers (the count order matters only there)
        keys = o.ft_table()[0]
        hot = [r for r in range(rs.n) if np.isin(o.kmer_hashes(r), keys).any()]
        assert len(hot) >= 3
        rids = hot[:10] + rids[:6]
    n = 0
    for rid in rids:
        r = _cmp(o, rs, rid, use_ft, bits)
        if r is None:
            continue
        got, exp = r
        assert got.shape == exp.shape and (got == exp).all(), (name, rid, got.shape, exp.shape)
        n += 1
    assert n > 5
