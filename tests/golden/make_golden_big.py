"""Golden fixture for a FULL-SIZE bench workload (default: BASELINE.json configs[2], chr1_250M_hifi30x = 250 Mb genome, 30x, 500 000
reads of 15 kb) from the REAL reference (oracle/_ref/ref_harness = unmodified hifiasm 0.25.0-r726 compiled from /root/reference).

Run in the build container only (needs /root/reference, ~20 GB of RAM, ~8 GB under /tmp, 15-40 min on 8 cores):
    python tests/golden/make_golden_big.py [workload] [n_sample] [bf_shift]
With bf_shift > 0 (the reference's own default is -f37) the run goes through the blocked Bloom filter; the fixture is then named
<workload>_f<bf_shift>.npz and also holds the all-k-mer histogram of ha_ft_gen and the filter table (--ft-tables).
What is stored (small enough for git):
  * the totals of the all-reads pass (overlaps, chained hits, hom_cov / het_cov, max_n_chain, occurrence thresholds),
  * the minimizer count histogram of ha_pt_gen,
  * a digest of EVERY read's result (ol, fake cigars, cl) and of every read's seed hits, folded over blocks of 256 reads
    (the digest is hao_batch_digest's, include/hao.h; the device computes it for the whole pass, so all 500 000 reads are compared),
  * for a sample of reads (default 256, evenly spread): the minimizers, ol->list and fake cigars verbatim + their per-read digests.
Reads are NOT stored: hifiasm_amd.synth regenerates them (deterministic C generator); CRCs of the lengths and of the first packed
megabyte detect drift.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hifiasm_amd import synth  # noqa: E402
from hifiasm_amd.workloads import WORKLOADS, LEN_JIT, workload_reads  # noqa: E402
from helpers import fold_digests  # noqa: E402
import oracle_py  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "chr1_250M_hifi30x"
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    bf = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    assert os.path.exists(harness)
    g, cov, L, err, rr, ont = WORKLOADS[name]
    genome = synth.make_genome(g, seed=11, repeat_rich=rr)
    n_reads = max(1, int(round(g * cov / L)))
    d = tempfile.mkdtemp(prefix="hao_goldbig_", dir=os.environ.get("HAO_TMP", "/tmp"))
    fa = os.path.join(d, "r.fq" if ont else "r.fa")
    synth.write_fasta_stream(fa, genome, n_reads, L, err, seed=12, len_jit=LEN_JIT.get(name, 0), fastq=bool(ont))
    step = max(1, n_reads // n_sample)
    sample = np.arange(step // 2, n_reads, step, dtype=np.uint64)[:n_sample]
    with open(os.path.join(d, "list.txt"), "w") as fp:
        fp.write("\n".join(str(int(x)) for x in sample) + "\n")
    cmd = [harness, "-t", str(os.cpu_count() or 8), "--time", "--dump", os.path.join(d, "s"), "--reads-list", os.path.join(d, "list.txt"),
           "--no-tables", "--nodump-hits", "--digest"]
    if ont:
        cmd.append("--ont")
    if bf:
        cmd += ["-f", str(bf), "--ft-tables"]
    cmd.append(fa)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    tj = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    pre = os.path.join(d, "s")
    ld = lambda nm, dt: np.fromfile(f"{pre}.{nm}", dtype=dt)  # noqa: E731
    meta = oracle_py.load_ref_meta(pre)
    dig = ld("dig.u64", np.uint64).reshape(-1, 2)
    assert dig.shape[0] == n_reads
    rs = workload_reads(name, want_codes=False)
    out = dict(
        sample=sample, mz_off=ld("mz_off.u64", np.uint64), mz=ld("mz.u64", np.uint64).reshape(-1, 2),
        ol_off=ld("ol_off.u64", np.uint64), ol=ld("ol.u32", np.uint32).reshape(-1, 12), fc_off=ld("fc_off.u64", np.uint64), fc=ld("fc.u64", np.uint64),
        pt_hist=ld("pt_hist.i64", np.int64), dig_sample=dig[sample.astype(np.int64)],
        dig_fold=fold_digests(dig[:, 0]), dig_kh_fold=fold_digests(dig[:, 1]),
        meta_keys=np.array(list(meta.keys()) + ["pass_overlaps", "pass_chained_hits"]),
        meta_vals=np.array(list(meta.values()) + [tj["overlaps"], tj["chained_hits"]], dtype=np.int64),
        len_crc=np.array([zlib.crc32(rs.lengths.tobytes()), zlib.crc32(rs.packed[: 1 << 20].tobytes())], dtype=np.uint64),
    )
    if bf:
        out.update(ft_hist=ld("ft_hist.i64", np.int64), ft_keys=ld("ft_keys.u64", np.uint64), ft_vals=ld("ft_vals.i32", np.int32))
        name = f"{name}_f{bf}"
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, json.dumps(tj), "->", os.path.getsize(path) // 1024, "KiB")
    with open(os.path.join(HERE, f"{name}.ref_time.json"), "w") as fp:      # the reference's own timing on this container's cores (informational)
        json.dump(dict(tj, host_cores=os.cpu_count(), note="build container, not the GPU box"), fp)
    shutil.rmtree(d)


if __name__ == "__main__":
    main()
