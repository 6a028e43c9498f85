"""The fuzzer's random workloads through the REAL reference: oracle/_ref/ref_harness (the unmodified hifiasm sources, oracle/Makefile) dumps its tables and every
read's minimizers, seed hits, ol->list, fake cigars and cl->list for a case's FASTA and options, and the C restatement (oracle/hao_oracle.c - what tests/test_gpu_fuzz.py
and the emulator sweeps compare the device library with) has to agree with all of it.  tests/ref_fuzz.py holds the plumbing.  Needs /root/reference's build
(the container that builds the repo): skipped where oracle/_ref is absent.

All 412 light cases agree (HAO_FUZZ_ALL=1 runs them: 115 ONT-mode cases through --rl-cut 0, 58 with a max_n_chain override through -N, 66 with --hg-size,
Bloom filters, HPC off, degenerate reads; 3.6 M overlaps, four minutes); the default run takes the 170 fastest of them (20 s).  The 68 repeat-dense cases are pinned the
other way round: tests/golden/fuzz_heavy.npz holds the REFERENCE's per-read digests (tests/golden/make_fuzz_golden.py), and test_heavy_digests_are_the_references
recomputes a few of them here."""
import os

import numpy as np
import pytest

import ref_fuzz
from helpers import GOLDEN

pytestmark = pytest.mark.skipif(not ref_fuzz.available(), reason="oracle/_ref/ref_harness not built (no /root/reference here)")

FAST = [12, 14, 18, 21, 23, 25, 29, 30, 32, 34, 36, 37, 38, 42, 45, 47, 49, 51, 53, 58, 61, 65, 67, 70, 83, 88, 89, 92, 94, 95, 101, 102, 108, 112, 113, 114, 116, 117, 118, 200, 202, 207,
        209, 211, 212, 213, 214, 219, 221, 222, 225, 228, 230, 231, 233, 234, 235, 239, 241, 242, 248, 250, 253, 256, 258, 261, 264, 266, 267, 269, 270, 272, 273, 275, 277, 279, 280, 281,
        283, 285, 286, 290, 293, 295, 296, 297, 301, 303, 305, 306, 309, 310, 311, 316, 318, 319, 320, 322, 323, 324, 326, 329, 332, 333, 339, 340, 348, 349, 353, 356, 358, 359, 361, 362,
        363, 366, 367, 371, 376, 377, 382, 383, 384, 386, 387, 388, 389, 391, 392, 395, 397, 1000, 1003, 1005, 1007, 1008, 1009, 1010, 1011, 1016, 1019, 1021, 1022, 1023, 1027, 1036, 1038,
        1040, 1043, 1044, 1047, 1049, 1050, 1057, 1058, 1066, 1068, 1073, 1081, 1089, 1097, 1099, 1100, 1114, 1115, 1126, 1140, 1145, 1146, 1154]


def _seeds():
    if os.environ.get("HAO_FUZZ_ALL"):
        import test_gpu_fuzz
        return list(test_gpu_fuzz.LIGHT_ALL)
    return FAST


@pytest.mark.parametrize("seed", _seeds())
def test_restatement_equals_the_reference(seed):
    n, tot, bad = ref_fuzz.compare(seed)
    assert not bad, (seed, bad)


def test_the_default_selection_covers_every_option_family():
    import simt_fuzz
    fam = {"ont": 0, "max_n_chain": 0, "hg_size": 0, "bf_shift": 0, "hpc0": 0, "degenerate": 0, "bw_thres": 0, "n_rate": 0}
    for s in FAST:
        d, o = simt_fuzz.case(s)
        fam["ont"] += bool(o.get("is_ont")); fam["max_n_chain"] += "max_n_chain" in o; fam["hg_size"] += "hg_size" in o; fam["bf_shift"] += "bf_shift" in o
        fam["hpc0"] += o.get("hpc", 1) == 0; fam["degenerate"] += "degenerate" in d; fam["bw_thres"] += "bw_thres" in o; fam["n_rate"] += "n_rate" in d
    assert all(v >= 10 for v in fam.values()), fam


@pytest.mark.parametrize("seed", [4, 27, 48, 1029, 1041])
def test_heavy_digests_are_the_references(seed):
    """tests/golden/fuzz_heavy.npz (what the repeat-dense cases of tests/test_gpu_fuzz.py are compared with) = ref_harness --digest, recomputed for five cases"""
    z = np.load(os.path.join(GOLDEN, "fuzz_heavy.npz"))
    g = ref_fuzz.digests(seed, threads=4)
    for k in ("peaks", "sketch", "hits", "result"):
        assert np.array_equal(z[f"s{seed}_{k}"], g[k]), (seed, k)
