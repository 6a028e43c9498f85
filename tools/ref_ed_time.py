"""The reference's own window edit distance (ed_band_cal_semi_64_w_absent_diag incl. recover_UC_Read_sub_region, as Correct.cpp:3897 calls it per candidate) on the
task set tools/bench_ed.py gives the device, on all host cores: oracle/_ref/ref_harness --ed-tasks --time -t N.  Build container only (needs /root/reference's build).
usage: ref_ed_time.py [scenario] [n_reads] [threads]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hifiasm_amd import synth
from helpers import ed_tasks, scenario_reads
name = sys.argv[1] if len(sys.argv) > 1 else "hifi_15k"; nr = int(sys.argv[2]) if len(sys.argv) > 2 else 400; T = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
rs, okw = scenario_reads(name)
t = ed_tasks(name, n_reads=nr, seed=11)
t = np.concatenate([t] * max(1, 400000 // max(1, t.shape[0])))
d = tempfile.mkdtemp(prefix="hao_edt_"); ont = bool(okw.get("is_ont")); fa = os.path.join(d, "r.fq" if ont else "r.fa")
synth.write_fasta(fa, rs, fastq=ont); t.tofile(os.path.join(d, "tasks.u32"))
r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_harness"), "-t", str(T), "--time", "--dump", os.path.join(d, "s"), "--reads-list", "/dev/null", "--no-tables",
                    "--ed-tasks", os.path.join(d, "tasks.u32")] + (["--ont"] if ont else []) + [fa], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-2000:]
j = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"ed_pairs"')][-1]
print(json.dumps(dict(j, scenario=name, text_bases=int(t[:, 6].sum()), host_cores=os.cpu_count())))
