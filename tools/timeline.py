"""Timeline of ONE delivered step out of a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py: where the device's main queue idles and what the copies run under.
usage: python tools/timeline.py <rocprofv3 output dir> [out.txt]
The step = from the last sketch_unit_kernel launch (ha_pt_gen starts a step) to the last event of the trace."""
import csv, glob, os, sys, collections

d = sys.argv[1]
kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
mf = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
K = []
with open(kf) as f:
    for r in csv.DictReader(f):
        K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
C = []
if mf:
    with open(mf[0]) as f:
        for r in csv.DictReader(f):
            C.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), r.get("Stream_Id", "?")))
K.sort(); C.sort()
t0 = max(k[0] for k in K if k[2].startswith("sketch_unit_kernel"))
K = [k for k in K if k[0] >= t0]; C = [c for c in C if c[1] >= t0]
t1 = max(max(k[1] for k in K), max([c[1] for c in C] or [0]))
out = []
P = out.append
P(f"step window {(t1 - t0) / 1e6:.2f} ms, {len(K)} kernels, {len(C)} copies")
bys = collections.defaultdict(list)
for k in K:
    bys[(k[3], k[4])].append(k)
for s, ks in sorted(bys.items(), key=lambda kv: -sum(k[1] - k[0] for k in kv[1])):
    P(f"queue/stream {s}: {len(ks)} kernels, busy {sum(k[1] - k[0] for k in ks) / 1e6:.2f} ms, first +{(ks[0][0] - t0) / 1e6:.2f} last +{(ks[-1][1] - t0) / 1e6:.2f}")
# union of all kernel intervals = device busy; the gaps
ev = sorted((k[0], k[1], k[2]) for k in K)
busy = 0; cur_s, cur_e, last_name = ev[0][0], ev[0][1], ev[0][2]; gaps = []
for s, e, n in ev[1:]:
    if s > cur_e:
        gaps.append((s - cur_e, cur_e - t0, last_name, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e; last_name = n
    elif e > cur_e:
        cur_e = e; last_name = n
busy += cur_e - cur_s
P(f"device busy (any kernel) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps; after the last kernel {(t1 - cur_e) / 1e6:.2f} ms")
hist = collections.Counter()
for g in gaps:
    hist[min(9, int(g[0] / 1e4).bit_length())] += g[0]
P("idle by gap size (bucket = 10 us x 2^k): " + ", ".join(f"<{10 * 2 ** k} us: {v / 1e6:.2f} ms" for k, v in sorted(hist.items())))
agg = collections.defaultdict(lambda: [0, 0])
for g in gaps:
    a = agg[(g[2], g[3])]; a[0] += g[0]; a[1] += 1
P("idle by (kernel before -> kernel after), top 25:")
for (a, b), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    P(f"  {t / 1e6:7.3f} ms in {n:4d} gaps   {a} -> {b}")
P("copies: " + ", ".join(f"{dr}: {sum(c[1] - c[0] for c in C if c[2] == dr) / 1e6:.2f} ms in {sum(1 for c in C if c[2] == dr)}" for dr in sorted({c[2] for c in C})))
big = [c for c in C if c[1] - c[0] > 1e6]
for c in big:
    P(f"  copy {c[2]} +{(c[0] - t0) / 1e6:8.2f} .. +{(c[1] - t0) / 1e6:8.2f} ms ({(c[1] - c[0]) / 1e6:.2f} ms)")
# kernel time by name on the whole step
kn = collections.defaultdict(lambda: [0, 0])
for k in K:
    kn[k[2]][0] += k[1] - k[0]; kn[k[2]][1] += 1
P("kernels by time, top 16:")
for n, (t, c) in sorted(kn.items(), key=lambda kv: -kv[1][0])[:16]:
    P(f"  {t / 1e6:8.3f} ms {c:5d}  {n}")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
