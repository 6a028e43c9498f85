"""In-tree build of every native artefact (no JIT cache: the .so files travel with
the gpurun snapshot).

  hifiasm_amd/libhao.so        product: HIP kernels (gfx950) + C-ABI (include/hao.h)
  hifiasm_amd/libhaosynth.so   bench/test tooling: synthetic read generator
  oracle/liboracle.so          TEST ORACLE: plain-C restatement of the reference path
  oracle/_ref/ref_harness      TEST ORACLE: the unmodified reference, only when
                               /root/reference is present (this container)
  oracle/_ref/hifiasm_ref|hao  the reference executable, plain and with its seam served by
                               libhao.so (integration/hao_hifiasm_shim.cpp)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hifiasm_amd")
CSRC = os.path.join(PKG, "csrc")
ORACLE = os.path.join(ROOT, "oracle")
REF = "/root/reference"

HIP_SOURCES = ["hao_capi.hip", "hao_f3.hip"]      # two translation units that #include the kernel files (the second: f3's 36 window-alignment kernels); compiled side by side
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
OBJ_DIR = os.path.join(PKG, "_obj")      # object files (git-ignored, not needed on the GPU box)


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd, cwd=None):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=cwd)


def build_synth(force=False):
    out = os.path.join(PKG, "libhaosynth.so")
    src = os.path.join(CSRC, "hao_synth.c")
    if force or _newer(out, [src]):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-o", out, src, "-lpthread"])
    return out


def build_hip(force=False):
    out = os.path.join(PKG, "libhao.so")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "hao.h")]
    if force or _newer(out, deps):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        os.makedirs(OBJ_DIR, exist_ok=True)
        objs, procs = [], []
        for src in HIP_SOURCES:
            obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
            objs.append(obj)
            if force or _newer(obj, deps):
                cmd = [hipcc] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
                print("[build]", " ".join(cmd), flush=True)
                procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, pr in procs:
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-L/opt/rocm/lib", "-lrccl", "-lpthread"])
    return out


def build_oracle(force=False):
    out = os.path.join(ORACLE, "liboracle.so")
    deps = [os.path.join(ORACLE, "hao_oracle.c"), os.path.join(ORACLE, "hao_oracle.h"), os.path.join(ORACLE, "hao_oracle_ed.inc")]
    if force or _newer(out, deps):
        _run(["make", "-C", ORACLE, "liboracle.so"] + (["-B"] if force else []))
    return out


def build_ref(force=False):
    """The real reference as a checker/baseline binary; only where its sources exist."""
    if not os.path.isdir(REF):
        return None
    out = os.path.join(ORACLE, "_ref", "ref_harness")
    deps = [os.path.join(ORACLE, "ref_harness.cpp"), os.path.join(ORACLE, "ref_htab_dump.cpp")]
    if force or _newer(out, deps):
        _run(["make", "-C", ORACLE, "-j8", "ref"])
    # drop-in demonstration binaries: plain reference + reference with the seam served by libhao.so
    out2 = os.path.join(ORACLE, "_ref", "hifiasm_hao")
    deps2 = [os.path.join(ROOT, "integration", "hao_hifiasm_shim.cpp"), os.path.join(ROOT, "include", "hao.h"), os.path.join(PKG, "libhao.so")]
    if force or _newer(out2, deps2):
        _run(["make", "-C", ORACLE, "-j8", "hao-hifiasm"])
    return out


def build_all(force=False):
    build_synth(force)
    if os.path.exists(os.path.join(CSRC, HIP_SOURCES[0])):
        build_hip(force)
    build_oracle(force)
    build_ref(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
