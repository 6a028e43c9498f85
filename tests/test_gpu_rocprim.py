"""Pins a rocPRIM behaviour the sharded index build depends on: radix_sort_pairs over a BIT RANGE with begin_bit > 0 has been seen
to mis-sort inputs of 5 k - 200 k elements on this ROCm (both overloads), which is why hao_pt_run groups by a separate 16-bit owner key with
begin_bit = 0.  hao_selftest_rocprim sorts the same random keys both ways on the device and reports (mismatches of the begin_bit = 48 sort,
mismatches of the separate-key sort); the second MUST be 0 - the first is recorded so that a ROCm upgrade that fixes (or changes) the
behaviour shows up in the test log instead of silently changing nothing or something."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [5_000, 20_000, 200_000, 4_000_000])
def test_owner_grouping_sort(n):
    from hifiasm_amd.api import lib
    L = lib()
    L.hao_selftest_rocprim.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 2)()
    assert L.hao_selftest_rocprim(n, out) == 0
    print(f"[rocprim] n={n}: begin_bit=48 stable-sort mismatches {out[0]}, separate 16-bit key mismatches {out[1]}")
    assert out[1] == 0


def test_more_than_2_32_items():
    """The k-mer occurrences of BASELINE configs[2] are 5.6 G: a kernel launch of more than 2^32 work-items and rocprim::run_length_encode (whose size
    parameter is an `unsigned int`) both silently handle n mod 2^32 items.  The engine's launch shape and its run-length helper on 2^32 + 2^20 keys."""
    from hifiasm_amd.api import lib
    L = lib()
    L.hao_selftest_big.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
    n = (1 << 32) + (1 << 20)
    out = (C.c_uint64 * 3)()
    assert L.hao_selftest_big(n, out) == 0
    assert (out[0], out[1], out[2]) == (n >> 3, n, 0)


@pytest.mark.parametrize("n", [9_000_000, 40_000_000])      # (hao_pt_gen takes this path from 2^23 minimizers on: below, rocprim's bit-range sort is the one pinned above)
def test_index_sort_on_40_bits(n):
    """hao_pt_gen sorts (hash, read-order index) pairs on hash bits 24 .. 63 and repairs the 40-bit runs that hold several keys (hao_index.cuh): on keys built
    to have many such runs the result must equal the stable 64-bit sort, element for element (keys and carried indices)."""
    from hifiasm_amd.api import Engine, lib
    L = lib()
    L.hao_selftest_sortbits.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    e = Engine(0)
    out = (C.c_uint64 * 4)()
    assert L.hao_selftest_sortbits(e.h, n, out) == 0
    e.close()
    print(f"[sort40] n={n}: differences {out[0]}, runs rewritten {out[1]}, scratch elements {out[2]}, overflow {out[3]}")
    assert out[3] == 0 and 1000 < out[1] <= max(1 << 16, n >> 10)      # (beyond that many runs hao_pt_gen falls back to the 64-bit sort)
    assert out[0] == 0
