"""TEST INFRASTRUCTURE: runs the `-m gpu` parity tests on the CPU against tests/simt/_build/libhao_simt.so - the device library's own sources (hifiasm_amd/csrc)
compiled by g++ for an emulated gfx950 workgroup (tests/simt/hip/hip_runtime.h: one fiber per work-item, cross-lane operations and barriers as rendezvous; the
device-wide rocPRIM primitives and the HIP host API as plain sequential code).  `reexport(globals(), "test_gpu_x")` copies a GPU test module's tests and fixtures
into a CPU test module, drops the scenarios that are too big for an emulation, and points hifiasm_amd.api at the emulated library for the duration of the module.
The product never loads that library: hifiasm_amd/api.py knows only libhao.so."""
import importlib
import os

import pytest

import simt_build
from scenarios import BIG_SCENARIOS

BIG = set(BIG_SCENARIOS) | ({"rr", "bf24", "f37", "bw001rr", "fz3"} if not os.environ.get("HAO_SIMT_FULL") else set())      # + the repeat-rich sets: 40 - 130 s each on the emulator


def _is_big(v):
    if isinstance(v, str):
        return v in BIG
    if isinstance(v, (tuple, list)):
        return any(_is_big(x) for x in v)
    if hasattr(v, "values"):      # pytest.param(...)
        return any(_is_big(x) for x in v.values)
    return False


FULL = bool(os.environ.get("HAO_SIMT_FULL"))      # everything that is feasible on the emulator (about half an hour) instead of the default selection


def _filtered_marks(fn, drop, replace):
    marks = []
    for m in getattr(fn, "pytestmark", []):
        if m.name == "parametrize":
            names, values = m.args[0], list(m.args[1])
            if names in replace:
                values = list(replace[names])
            values = [v for v in values if not _is_big(v) and not drop(v)]
            marks.append(pytest.mark.parametrize(names, values, **m.kwargs).mark)
        else:
            marks.append(m)
    return marks


def reexport(ns, module_name, only=None, skip=(), drop=lambda v: False, replace=None, keep=None):
    """only / skip: test function names; drop(value) -> True removes a parametrize value or fixture param; replace: {argnames: values} for a parametrize mark or
    {fixture name: params}; keep: scenario names of the default selection (every parametrize value / fixture param that is a scenario name outside it is dropped
    unless HAO_SIMT_FULL is set)"""
    replace = replace or {}
    if keep is not None and not FULL:
        from scenarios import SCENARIOS
        drop0 = drop

        def outside(v):
            if isinstance(v, str):
                return v in SCENARIOS and v not in keep
            if isinstance(v, (tuple, list)):
                return any(outside(x) for x in v)
            return any(outside(x) for x in v.values) if hasattr(v, "values") else False
        drop = lambda v: drop0(v) or outside(v)
    mod = importlib.import_module(module_name)
    for name, obj in list(vars(mod).items()):
        if type(obj).__name__ == "FixtureFunctionDefinition":
            mk = obj._fixture_function_marker
            params = None if mk.params is None else [p for p in (replace.get(name) or mk.params) if not _is_big(p) and not drop(p)]
            ns[name] = pytest.fixture(scope=mk.scope, params=params, autouse=mk.autouse, ids=mk.ids, name=mk.name)(obj._get_wrapped_function())
        elif name.startswith("test_") and callable(obj):
            if (only and name not in only) or name in skip:
                continue
            import functools
            import types
            f = types.FunctionType(obj.__code__, obj.__globals__, name, obj.__defaults__, obj.__closure__)
            f.__doc__ = obj.__doc__; f.__kwdefaults__ = obj.__kwdefaults__
            f.pytestmark = _filtered_marks(obj, drop, replace)
            ns[name] = f

    @pytest.fixture(scope="module", autouse=True)
    def _simt_library():
        from hifiasm_amd import api
        old_path, old_lib = api.lib_path, api._LIB
        path = simt_build.build_lib()
        api.lib_path = lambda: path; api._LIB = None
        old_env = os.environ.get("HAO_SIMT_ZERO")
        yield
        api.lib_path, api._LIB = old_path, old_lib
    ns["_simt_library"] = _simt_library
