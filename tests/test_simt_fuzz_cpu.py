"""Random small workloads and option mixes (tests/simt_fuzz.py: k 11 - 63, w 3 - 99, HPC on / off, ONT mode, Bloom filters, band widths, --hg-size, max_n_chain, N bases,
repeat content, reads of 300 - 7000 bases) through the emulated device library against the oracle.  A fixed handful here; `python tests/simt_fuzz.py SEED ...` for more
(three sweeps - seeds 1 - 120, 200 - 399 and, with degenerate reads mixed in, 1000 - 1159 - found nothing)."""
import pytest

import simt_fuzz


@pytest.mark.parametrize("seed", [3, 14, 23, 25, 38, 42, 1003, 1008, 1011, 1012])
def test_random_workload(seed):
    from hifiasm_amd import api
    old = api.lib_path, api._LIB
    try:
        d, okw, n, tot, bad = simt_fuzz.run(seed)
    finally:
        api.lib_path, api._LIB = old
    print(f"[simt fuzz] seed {seed}: {n} reads, {tot} overlaps, {d} {okw}")
    assert not bad, (bad, d, okw)
