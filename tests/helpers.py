"""Shared helpers for the test-suite (tests only)."""
import os
import zlib

import numpy as np

from hifiasm_amd import synth
from scenarios import SCENARIOS, BIG_SCENARIOS, build_reads
import oracle_py

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_CACHE = {}


def scenario_reads(name):
    if name not in _CACHE:
        dkw, okw = (SCENARIOS.get(name) or BIG_SCENARIOS[name])
        _CACHE[name] = (build_reads(dkw), okw)
    return _CACHE[name]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = dict(zip([str(x) for x in d.pop("meta_keys")], [int(x) for x in d.pop("meta_vals")]))
    return d


_ORACLES = {}


def scenario_oracle(name):
    """Oracle with ft_gen + pt_gen already run (cached per session)."""
    if name not in _ORACLES:
        rs, okw = scenario_reads(name)
        o = oracle_py.Oracle(rs.codes, rs.code_off, **okw)
        o.ft_gen()
        o.pt_gen()
        _ORACLES[name] = o
    return _ORACLES[name]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())
