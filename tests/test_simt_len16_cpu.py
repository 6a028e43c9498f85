"""The last read length whose positions fit 16 bits, on the CPU emulation of the device library: reads cut to EXACTLY 65 535 bases ("len65535": ~1900 minimizers per read,
every read left to the table kernels; "len65535w": the same reads at w = 101, ~950 minimizers, i.e. the list-major seed kernel with its 6-byte LDS records) - seed hits,
h_ec_lchain and the delivered bytes (minimizer tables in their 4-byte form) against the oracle.  On the device the two scenarios ride along in tests/test_gpu_overlap.py and
tests/test_gpu_stream.py; tests/simt_suite.py keeps BIG_SCENARIOS out of the emulator's default selection, these two take 6 and 12 s."""
import pytest

import simt_build
from helpers import scenario_reads, scenario_oracle


@pytest.fixture(scope="module", autouse=True)
def _simt_library():
    from hifiasm_amd import api
    old_path, old_lib = api.lib_path, api._LIB
    path = simt_build.build_lib()
    api.lib_path = lambda: path; api._LIB = None
    yield
    api.lib_path, api._LIB = old_path, old_lib


def _same(a, b):
    return all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))


@pytest.mark.parametrize("name,left", [("len65535w", 0), ("len65535", None)])
def test_positions_up_to_65534(name, left):
    from hifiasm_amd.api import Engine
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    assert int(rs.lengths.max()) == 65535
    e = Engine(0, **okw)
    try:
        e.set_readset(rs); e.ha_ft_gen(); e.ha_pt_gen()
        e.overlap_batch(0, rs.n)
        sp = e.batch_seed_path()
        assert sp["first_launch"] == "seed_lds_kernel" and sp["left_to_tables"] == (rs.n if left is None else left), sp
        bad = [r for r in range(rs.n) if not ((e.fetch_seed_hits(r).shape == o.seed_hits(r).shape and (e.fetch_seed_hits(r) == o.seed_hits(r)).all()) and _same(e.h_ec_lchain(r), o.lchain(r)))]
        assert not bad, bad[:8]
        d = e.deliver_wait(e.overlap_batch_async(0, rs.n))
        assert d.qmz_pos and not d.qmz      # (every read shorter than 65 536 bases: the packed tables)
        bad = [r for r in range(rs.n) if not _same(e.delivered_read(d, r), o.lchain(r))]
        assert not bad, bad[:8]
    finally:
        e.close()
