"""Which device kernels has the CPU emulation (tests/simt) executed?  Runs the given pytest modules (default: every tests/test_simt_*_cpu.py) with HAO_SIMT_PROF=1,
collects the per-kernel launch counts the emulated library prints at exit, and sets them against the `__global__` functions defined in hifiasm_amd/csrc.
usage: [HAO_SIMT_FULL=1] python tools/simt_coverage.py [-n WORKERS] [modules ...] > profiles/rNN/simt_kernel_coverage.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
if "-h" in args or "--help" in args:
    print(__doc__); sys.exit(0)
workers = []
if args[:1] == ["-n"]:
    workers = ["-n", args[1]]; args = args[2:]
mods = args or sorted(glob.glob(os.path.join(ROOT, "tests", "test_simt_*_cpu.py")))
r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-s"] + workers + mods, capture_output=True, text=True, env=dict(os.environ, HAO_SIMT_PROF="1"), cwd=os.path.join(ROOT, "tests"))
tail = [l for l in r.stdout.splitlines() if " passed" in l or " failed" in l]
seen = {}
for l in (r.stdout + r.stderr).splitlines():
    m = re.match(r"\[simt prof\]\s+([\d.]+) s\s+(\d+) launches\s+(.*)$", l)
    if m and not m.group(3).startswith("total"):
        name = re.sub(r"^void ", "", m.group(3)).strip()
        base = name.split("<")[0]
        e = seen.setdefault(base, {"launches": 0, "seconds": 0.0, "instances": set()})
        e["launches"] += int(m.group(2)); e["seconds"] += float(m.group(1)); e["instances"].add(name)
defined = {}
for f in sorted(glob.glob(os.path.join(ROOT, "hifiasm_amd", "csrc", "*"))):
    txt = re.sub(r"__launch_bounds__\([^)]*\)|__attribute__\(\([^)]*\)\)\)?", " ", open(f, errors="ignore").read())
    for m in re.finditer(r"__global__[^;{(]*?\b(\w+)\s*\(", txt):
        defined.setdefault(m.group(1), os.path.basename(f))
print(f"# kernels executed by the CPU emulation ({'HAO_SIMT_FULL=1, ' if os.environ.get('HAO_SIMT_FULL') else ''}{len(mods)} modules; pytest: {'; '.join(tail)})")
print(f"# {sum(1 for k in defined if k in seen)} of {len(defined)} __global__ functions of hifiasm_amd/csrc were launched\n")
for k in sorted(defined, key=lambda k: (k not in seen, defined[k], k)):
    if k in seen:
        e = seen[k]
        print(f"{k:34s} {defined[k]:18s} {e['launches']:8d} launches {e['seconds']:9.1f} s  {len(e['instances'])} instance(s): {', '.join(sorted(e['instances']))[:200]}")
    else:
        print(f"{k:34s} {defined[k]:18s} NOT LAUNCHED")
