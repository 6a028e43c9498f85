# Throughput of the window-alignment kernels (f3) on window / candidate pairs formed like Correct.cpp:3897 does, next to the C restatement on one host core.
# All five modes; per mode the device time is split into the whole call (task upload, sort, kernels, result download) and the kernels alone
# (hao_stage_times is not wired for f3, so the kernel share comes from `rocprofv3 --kernel-trace --stats -- python tools/bench_ed.py`).
# (The reference's own function on all host cores: tools/ref_ed_time.py, build container only.)
# usage: bench_ed.py [scenario] [n_reads] [--wide 0|1|2] [--dry]      (--dry: build the task sets and the CPU side only - runs without a GPU)
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # resolved from this file: the tool may be started from any directory (rocprofv3 runs it from /tmp)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import ed_tasks, ed_tasks_grid, ed_global_tasks, ed_semi_trace_tasks, ed_ext_tasks, scenario_reads, scenario_oracle  # noqa: E402

import argparse  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("scenario", nargs="?", default="hifi_15k"); ap.add_argument("n_reads", nargs="?", type=int, default=400)
ap.add_argument("--wide", type=int, default=0, help="0: thre <= 31 (one-word bands), 1: 32 .. 63 (two words), 2: 64 .. 127 (three / four words)")
ap.add_argument("--dry", action="store_true")
ap.add_argument("--grid", action="store_true", help="distance-only mode on the reference's fixed window grid: every overlap of a read contributes a pair per 375-base window it covers (the candidates of a window share their text); without it: the test generator's windows, anchored at each overlap's start (hardly any two pairs share a text)")
a_ = ap.parse_args()
name, nr, wide, dry = a_.scenario, a_.n_reads, a_.wide, a_.dry
rs, okw = scenario_reads(name)
o = scenario_oracle(name)
e = None
if not dry:
    from hifiasm_amd.api import Engine
    e = Engine(0, **okw); e.set_readset(rs)
CAP = 80 if wide == 0 else 264
MODES = (("semi, distance only", lambda: (ed_tasks_grid if a_.grid else ed_tasks)(name, n_reads=nr, seed=11, wide=wide), lambda t: e.window_ed_batch(t), lambda t: o.window_ed(t)),
         ("global + cigar", lambda: ed_global_tasks(name, n_reads=nr, seed=12, wide=wide), lambda t: e.window_trace_batch(t, cap=CAP), lambda t: o.window_trace(t, cap=CAP)),
         ("extension fwd + cigar", lambda: ed_ext_tasks(name, n_reads=nr, seed=14, wide=wide), lambda t: e.window_trace_batch(t, cap=CAP, mode=1), lambda t: o.window_trace(t, cap=CAP, mode=1)),
         ("extension bwd + cigar", lambda: ed_ext_tasks(name, n_reads=nr, seed=14, wide=wide), lambda t: e.window_trace_batch(t, cap=CAP, mode=2), lambda t: o.window_trace(t, cap=CAP, mode=2)),
         ("semi + cigar", lambda: ed_semi_trace_tasks(name, n_reads=nr, seed=13, wide=wide), lambda t: e.window_trace_batch(t, cap=CAP, mode=3), lambda t: o.window_trace(t, cap=CAP, mode=3)))
for label, mk, gpu, cpu in (MODES[:1] if a_.grid else MODES):
    t = mk()
    if not a_.grid:
        t = np.concatenate([t] * max(1, 400000 // max(1, t.shape[0])))
    if a_.grid:      # how many pairs share a text (= lanes of a wave that step together)
        _, cnt_ = np.unique(t[:, 4:8], axis=0, return_counts=True)
        print(f"[grid] {t.shape[0]} pairs over {cnt_.size} distinct texts: {t.shape[0] / cnt_.size:.1f} pairs per text on average", flush=True)
    bases = int(t[:, 6].sum())
    tg = None
    if not dry:
        gpu(t)                                                    # (first call: the engine allocates its scratch buffers; they are kept between calls)
        best = 1e9
        for _ in range(3):
            t0 = time.time(); gpu(t); best = min(best, time.time() - t0)
        tg = best
    sub = t[: min(t.shape[0], 40000)]
    tc = None
    if wide < 2:                                                  # (the C restatement covers bands of one and two words)
        t0 = time.time(); cpu(sub); tc = (time.time() - t0) * t.shape[0] / sub.shape[0]
    msg = f"{label:22s} {t.shape[0]} pairs, {bases / 1e6:.0f} M text bases"
    if tg is not None:
        msg += f": device call (task upload + sort + kernels + result download) {tg * 1e3:.1f} ms = {t.shape[0] / tg / 1e6:.2f} M pairs/s"
    if tc is not None:
        msg += f"; C restatement, one core {tc * 1e3:.0f} ms = {t.shape[0] / tc / 1e6:.3f} M pairs/s" + (f"; ratio {tc / tg:.0f}" if tg else "")
    print(msg, flush=True)
if e is not None:
    e.close()
