# Throughput of the window-alignment kernels (f3) on window / candidate pairs formed like Correct.cpp:3897 does, next to the C restatement on one host core.
# (The reference's own function on all host cores: tools/ref_ed_time.py, build container only.)
# usage: bench_ed.py [scenario] [n_reads]
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import ed_tasks, ed_global_tasks, ed_semi_trace_tasks, scenario_reads, scenario_oracle
from hifiasm_amd.api import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "hifi_15k"
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rs, okw = scenario_reads(name)
o = scenario_oracle(name)
e = Engine(0, **okw); e.set_readset(rs)
for label, t, gpu, cpu in (("semi, distance only", ed_tasks(name, n_reads=nr, seed=11), lambda t: e.window_ed_batch(t), lambda t: o.window_ed(t)),
                           ("global + cigar", ed_global_tasks(name, n_reads=nr, seed=12), lambda t: e.window_trace_batch(t), lambda t: o.window_trace(t)),
                           ("semi + cigar", ed_semi_trace_tasks(name, n_reads=nr, seed=13), lambda t: e.window_trace_batch(t, mode=3), lambda t: o.window_trace(t, mode=3))):
    t = np.concatenate([t] * max(1, 400000 // max(1, t.shape[0])))
    bases = int(t[:, 6].sum())
    gpu(t); t0 = time.time(); gpu(t); tg = time.time() - t0      # (second call: the engine keeps its scratch buffers between calls)
    sub = t[: min(t.shape[0], 40000)]; t0 = time.time(); cpu(sub); tc = (time.time() - t0) * t.shape[0] / sub.shape[0]
    print(f"{label:22s} {t.shape[0]} pairs, {bases / 1e6:.0f} M text bases: device (incl. task upload / result download) {tg * 1e3:.1f} ms = {t.shape[0] / tg / 1e6:.2f} M pairs/s; "
          f"C restatement, one core {tc * 1e3:.0f} ms = {t.shape[0] / tc / 1e6:.3f} M pairs/s; ratio {tc / tg:.0f}")
