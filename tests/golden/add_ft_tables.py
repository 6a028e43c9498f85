"""Adds the all-k-mer histogram and the filter table of ha_ft_gen (ft_hist, ft_keys, ft_vals) to an existing full-size -f0 fixture
(tests/golden/<workload>.npz) from the REAL reference (oracle/_ref/ref_harness --ft-tables).  Build container only; ~10 min on 8 cores for
chr1_250M_hifi30x.    python tests/golden/add_ft_tables.py [workload]"""
import json, os, shutil, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hifiasm_amd import synth  # noqa: E402
from hifiasm_amd.workloads import WORKLOADS  # noqa: E402
import oracle_py  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "chr1_250M_hifi30x"
harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
g, cov, L, err, rr, ont = WORKLOADS[name]
genome = synth.make_genome(g, seed=11, repeat_rich=rr)
n_reads = max(1, int(round(g * cov / L)))
d = tempfile.mkdtemp(prefix="hao_goldft_", dir=os.environ.get("HAO_TMP", "/tmp"))
fa = os.path.join(d, "r.fq" if ont else "r.fa")
synth.write_fasta_stream(fa, genome, n_reads, L, err, seed=12, fastq=bool(ont))
with open(os.path.join(d, "list.txt"), "w") as fp:
    fp.write("0\n")
cmd = [harness, "-t", str(os.cpu_count() or 8), "--time", "--dump", os.path.join(d, "s"), "--reads-list", os.path.join(d, "list.txt"), "--no-tables", "--nodump-hits", "--ft-tables"]
if ont:
    cmd.append("--ont")
r = subprocess.run(cmd + [fa], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-3000:]
pre = os.path.join(d, "s")
path = os.path.join(HERE, f"{name}.npz")
old = dict(np.load(path, allow_pickle=False))
meta = oracle_py.load_ref_meta(pre)
om = dict(zip([str(k) for k in old["meta_keys"]], [int(v) for v in old["meta_vals"]]))
assert all(om[k] == v for k, v in meta.items() if k in om and k not in ("tot_ol", "tot_cl", "ft_distinct")), "the reference run differs from the one behind the fixture"
old.update(ft_hist=np.fromfile(pre + ".ft_hist.i64", dtype=np.int64), ft_keys=np.fromfile(pre + ".ft_keys.u64", dtype=np.uint64), ft_vals=np.fromfile(pre + ".ft_vals.i32", dtype=np.int32))
np.savez_compressed(path, **old)
print(name, "ft_hist sum", int(old["ft_hist"].sum()), "ft keys", old["ft_keys"].size)
shutil.rmtree(d)
