"""TEST INFRASTRUCTURE: builds tests/simt/_build/libsimt_*.so - the device sources of hifiasm_amd/csrc compiled by g++ against the emulated-workgroup stand-in for
<hip/hip_runtime.h> (tests/simt/hip/hip_runtime.h).  The sources are used as they are, except for one mechanical rewrite g++ needs: the declaration of the
dynamic LDS array (`extern __shared__ T name[];`) becomes a pointer to the emulator's LDS block."""
import contextlib
import fcntl
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hifiasm_amd", "csrc")
SIMT = os.path.join(ROOT, "tests", "simt")
BUILD = os.path.join(SIMT, "_build")
_DYN = re.compile(r"extern\s+__shared__\s+(?:__align__\(\d+\)\s+)?([\w ]+?)\s+(\w+)\s*\[\s*\]\s*;")


# clang's vector types (whole-vector loads / stores and .x are all the sources use)
_VEC4 = re.compile(r"typedef uint32_t (\w+) __attribute__\(\(ext_vector_type\(4\)\)\);")
# the one inline-assembly statement of the device sources (hao_sketch3.cuh: canonical strand = the borrow of f1 - r1, then four selects) in C++
_ASM = re.compile(r'asm\("v_sub_co_u32 %4, vcc, %5, %9.*?: "vcc"\);', re.S)
_ASM_CPP = ("{ const bool fw_ = ((uint64_t)f1h << 32 | f1l) < ((uint64_t)r1h << 32 | r1l); x0l = fw_ ? f0l : r0l; x0h = fw_ ? f0h : r0h; "
            "x1l = fw_ ? f1l : r1l; x1h = fw_ ? f1h : r1h; tmp = 0; }")


@contextlib.contextmanager
def _locked():
    """one builder at a time (test processes running side by side would otherwise compile into the same files)"""
    os.makedirs(BUILD, exist_ok=True)
    with open(os.path.join(BUILD, ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _patched_sources():
    os.makedirs(BUILD, exist_ok=True)
    newest = 0.0
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".cuh", ".hpp", ".hip")):
            continue
        src = open(os.path.join(CSRC, f)).read()
        out = _DYN.sub(lambda m: f"{m.group(1)} *{m.group(2)} = ({m.group(1)}*)hao_simt::dyn_lds();", src)
        out = _VEC4.sub(lambda m: f"struct {m.group(1)} {{ uint32_t x, y, z, w; }};", out)
        if f == "hao_sketch3.cuh":
            out, k = _ASM.subn(_ASM_CPP, out)
            assert k == 1, "hao_sketch3.cuh: the inline assembly statement this build rewrites has changed"
        dst = os.path.join(BUILD, f)
        if not os.path.exists(dst) or open(dst).read() != out:
            open(dst, "w").write(out)
        newest = max(newest, os.path.getmtime(dst))
    return newest


def build(name):
    """name: 'seed' -> tests/simt/seed_harness.cpp -> _build/libsimt_seed.so; returns the path"""
    with _locked():
        return _build(name)


def _build(name):
    newest = _patched_sources()
    src = os.path.join(SIMT, f"{name}_harness.cpp")
    shim = os.path.join(SIMT, "hip", "hip_runtime.h")
    out = os.path.join(BUILD, f"libsimt_{name}.so")
    newest = max(newest, os.path.getmtime(src), os.path.getmtime(shim))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        cmd = ["g++", "-O1", "-g", "-rdynamic", "-std=c++17", "-shared", "-fPIC", "-fno-strict-aliasing", "-w", "-I", SIMT, "-I", BUILD, "-I", os.path.join(ROOT, "include"), src, "-o", out + ".tmp"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("g++ failed:\n" + r.stderr[-6000:])
        os.replace(out + ".tmp", out)
    return out


def build_lib():
    """the whole device library (hao_capi.hip + hao_f3.hip, the C-ABI of include/hao.h) for the emulated workgroup -> _build/libhao_simt.so"""
    with _locked():
        return _build_lib()


def _build_lib():
    newest = _patched_sources()
    shims = [os.path.join(SIMT, "hip", "hip_runtime.h"), os.path.join(SIMT, "rocprim", "rocprim.hpp"), os.path.join(SIMT, "rccl", "rccl.h"), os.path.join(ROOT, "include", "hao.h")]
    newest = max([newest] + [os.path.getmtime(x) for x in shims])
    out = os.path.join(BUILD, "libhao_simt.so")
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        objs, procs = [], []
        for tu in ("hao_capi.hip", "hao_f3.hip"):
            obj = os.path.join(BUILD, tu.replace(".hip", ".simt.o")); objs.append(obj)
            cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-w", "-x", "c++", "-I", SIMT, "-I", BUILD, "-I", os.path.join(ROOT, "include"),
                   "-c", os.path.join(BUILD, tu), "-o", obj]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        for cmd, pr in procs:
            so, se = pr.communicate()
            if pr.returncode:
                raise RuntimeError("g++ failed: " + " ".join(cmd) + "\n" + se[-8000:])
        r = subprocess.run(["g++", "-shared", "-rdynamic", "-o", out + ".tmp"] + objs + ["-lpthread", "-ldl"], capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("link failed:\n" + r.stderr[-6000:])
        os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build("seed"))
    print(build_lib())
