"""TEST INFRASTRUCTURE: builds tests/simt/_build/libsimt_*.so - the device sources of hifiasm_amd/csrc compiled by g++ against the emulated-workgroup stand-in for
<hip/hip_runtime.h> (tests/simt/hip/hip_runtime.h).  The sources are used as they are, except for one mechanical rewrite g++ needs: the declaration of the
dynamic LDS array (`extern __shared__ T name[];`) becomes a pointer to the emulator's LDS block."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hifiasm_amd", "csrc")
SIMT = os.path.join(ROOT, "tests", "simt")
BUILD = os.path.join(SIMT, "_build")
_DYN = re.compile(r"extern\s+__shared__\s+(?:__align__\(\d+\)\s+)?([\w ]+?)\s+(\w+)\s*\[\s*\]\s*;")


def _patched_sources():
    os.makedirs(BUILD, exist_ok=True)
    newest = 0.0
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".cuh"):
            continue
        src = open(os.path.join(CSRC, f)).read()
        out = _DYN.sub(lambda m: f"{m.group(1)} *{m.group(2)} = ({m.group(1)}*)hao_simt::dyn_lds();", src)
        dst = os.path.join(BUILD, f)
        if not os.path.exists(dst) or open(dst).read() != out:
            open(dst, "w").write(out)
        newest = max(newest, os.path.getmtime(dst))
    return newest


def build(name):
    """name: 'seed' -> tests/simt/seed_harness.cpp -> _build/libsimt_seed.so; returns the path"""
    newest = _patched_sources()
    src = os.path.join(SIMT, f"{name}_harness.cpp")
    shim = os.path.join(SIMT, "hip", "hip_runtime.h")
    out = os.path.join(BUILD, f"libsimt_{name}.so")
    newest = max(newest, os.path.getmtime(src), os.path.getmtime(shim))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        cmd = ["g++", "-O2", "-g", "-rdynamic", "-std=c++17", "-shared", "-fPIC", "-fno-strict-aliasing", "-w", "-I", SIMT, "-I", BUILD, "-I", os.path.join(ROOT, "include"), src, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("g++ failed:\n" + r.stderr[-6000:])
    return out
