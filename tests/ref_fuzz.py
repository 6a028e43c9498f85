"""TEST INFRASTRUCTURE: the random workloads of tests/simt_fuzz.py through the REAL reference (oracle/_ref/ref_harness = the unmodified hifiasm sources compiled
from /root/reference by oracle/Makefile) - what pins the C restatement (oracle/hao_oracle.c), and through it the device library, on the fuzzer's option mixes.

  run_ref(seed, ...)        writes the case's reads as FASTA / FASTQ, runs ref_harness --dump with the case's options, returns the dump (oracle_py.load_ref_dump)
  compare(seed)             the dump against oracle_py.Oracle: coverage peaks, the position index, every read's minimizers, seed hits, ol->list, fake cigars, cl->list
  digests(seed)             per-read digests from ref_harness --digest (the repeat-dense cases: tests/golden/make_fuzz_golden.py)

The reference's option parser gets the case's options unchanged (-k -w -f -N --ont --hg-size --rl-cut --sc-cut; --no-hpc and --bw are the harness's own switches for
HA_F_NO_HPC and the pass's bw_thres).  Two adaptations, both on the ORACLE's side so that the two programs see the same input: --hg-size only parses k / m / g
suffixed sizes (CommandLines.cpp:848-863), so a case's hg_size is rounded down to a multiple of 1000 for both; ONT cases run with --rl-cut 0 --sc-cut 0 (the reader
otherwise drops reads shorter than 1000, htab.cpp:763, and the fuzzer's reads are 300 - 7000 bases)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")


def available():
    return os.path.exists(HARNESS)


def case(seed):
    """the case's read set and options as BOTH programs get them"""
    import simt_fuzz
    rs, d, okw = simt_fuzz.reads_of(seed)
    okw = dict(okw)
    if "hg_size" in okw:
        okw["hg_size"] = max(1000, okw["hg_size"] // 1000 * 1000)
    return rs, d, okw


def harness_args(okw, threads=2):
    cmd = [HARNESS, "-t", str(threads)]
    ont = bool(okw.get("is_ont"))
    if ont:
        cmd += ["--ont", "--rl-cut", "0", "--sc-cut", "0"]
    if "k" in okw:
        cmd += ["-k", str(okw["k"])]
    if "w" in okw:
        cmd += ["-w", str(okw["w"])]
    if "bf_shift" in okw:
        cmd += ["-f", str(okw["bf_shift"])]
    if okw.get("hpc", 1) == 0:
        cmd.append("--no-hpc")
    if "bw_thres" in okw:
        cmd += ["--bw", repr(okw["bw_thres"])]
    if "hg_size" in okw:
        cmd += ["--hg-size", f"{okw['hg_size'] // 1000}k"]
    if "max_n_chain" in okw:
        cmd += ["-N", str(okw["max_n_chain"])]
    return cmd, ont


def run_ref(seed, extra=(), threads=2, keep=None):
    """-> (rs, d, okw, dump dict, stderr tail); extra: more harness switches (e.g. ("--digest", "--no-tables", "--nodump-hits"))"""
    from hifiasm_amd import synth
    import oracle_py
    rs, d, okw = case(seed)
    tmp = keep or tempfile.mkdtemp(prefix="hao_reffuzz_")
    try:
        cmd, ont = harness_args(okw, threads)
        fa = os.path.join(tmp, "r.fq" if ont else "r.fa")
        synth.write_fasta(fa, rs, fastq=ont)
        cmd += ["--dump", os.path.join(tmp, "s")] + list(extra) + [fa]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"ref_harness failed on seed {seed}: {' '.join(cmd)}\n{r.stderr[-1500:]}")
        dump = oracle_py.load_ref_dump(os.path.join(tmp, "s"))
        return rs, d, okw, dump, r.stderr[-300:]
    finally:
        if keep is None:
            shutil.rmtree(tmp, ignore_errors=True)


def compare(seed):
    """-> (n reads, overlaps, list of differences)"""
    import oracle_py
    rs, d, okw, g, _ = run_ref(seed)
    M = g["meta"]
    bad = []
    if M["n_reads"] != rs.n or not np.array_equal(g["rlen"], rs.lengths):
        return rs.n, 0, [f"the reference kept {M['n_reads']} of {rs.n} reads"]
    o = oracle_py.Oracle(rs.codes, rs.code_off, **okw)
    ft = o.ft_gen()
    if ft != M["hom_cov_ft"]:
        bad.append(f"ft peak {ft} != {M['hom_cov_ft']}")
    hom = o.pt_gen(); st = o.stats()
    for key in ("hom_cov", "het_cov", "max_n_chain", "high_occ", "low_occ"):
        if st[key] != M[key]:
            bad.append(f"{key} {st[key]} != {M[key]}")
    if not np.array_equal(o.ft_hist(), g["ft_hist"]):
        bad.append("ft_hist")
    if not np.array_equal(o.pt_hist(), g["pt_hist"]):
        bad.append("pt_hist")
    pk, po, pp = o.pt_table()
    if not (np.array_equal(pk, g["pt_keys"]) and np.array_equal(po, g["pt_off"]) and np.array_equal(pp, g["pt_pos"])):
        bad.append("pt table")
    mz, kh, ol_all, cl_all, fc_all = g["mz"].reshape(-1, 2), g["kh"].reshape(-1, 4), g["ol"].reshape(-1, 12), g["cl"].reshape(-1, 4), g["fc"]
    n_mz = n_kh = n_ol = n_cl = 0
    for r in range(rs.n):
        a = o.sketch(r); b = mz[int(g["mz_off"][r]):int(g["mz_off"][r + 1])]
        n_mz += int(a.shape != b.shape or (a != b).any())
        a = o.seed_hits(r); b = kh[int(g["kh_off"][r]):int(g["kh_off"][r + 1])]
        n_kh += int(a.shape != b.shape or (a != b).any())
        ol, fc, fo, cl = o.lchain(r)
        s, e = int(g["ol_off"][r]), int(g["ol_off"][r + 1])
        rf = fc_all[int(g["fc_off"][s]):int(g["fc_off"][e])]
        n_ol += int(ol.shape != ol_all[s:e].shape or (ol != ol_all[s:e]).any() or fc.shape != rf.shape or (fc != rf).any())
        b = cl_all[int(g["cl_off"][r]):int(g["cl_off"][r + 1])]
        n_cl += int(cl.shape != b.shape or (cl != b).any())
    for nm, v in (("minimizers", n_mz), ("seed hits", n_kh), ("ol / fake cigars", n_ol), ("cl", n_cl)):
        if v:
            bad.append(f"{nm} of {v} reads")
    return rs.n, int(M["tot_ol"]), bad


def digests(seed, threads=8):
    """the per-read digests of the case from the reference (ref_harness --digest: [digest of (ol, fc, cl), digest of the seed hits] per read, all threads), the
    crc32 of every read's minimizers and the peaks: the arrays of tests/golden/fuzz_heavy.npz"""
    from helpers import crc
    rs, d, okw, g, _ = run_ref(seed, extra=("--digest", "--no-tables", "--nodump-hits", "--time"), threads=threads)
    M = g["meta"]
    assert M["n_reads"] == rs.n, (seed, M["n_reads"], rs.n)
    mz = g["mz"].reshape(-1, 2)
    sk = np.array([crc(mz[int(g["mz_off"][r]):int(g["mz_off"][r + 1])]) for r in range(rs.n)], dtype=np.uint32)
    dig = g["dig"].reshape(-1, 2)
    return dict(peaks=np.array([M["hom_cov_ft"], M["hom_cov"], M["het_cov"], rs.n, M["tot_ol"]], dtype=np.int64), sketch=sk, hits=dig[:, 1].copy(), result=dig[:, 0].copy())


if __name__ == "__main__":
    import sys
    import time
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    for s in sys.argv[1:]:
        t0 = time.time()
        n, tot, bad = compare(int(s))
        print(f"seed {s}: {'OK' if not bad else 'DIFF ' + '; '.join(bad)}  reads {n} overlaps {tot}  {time.time() - t0:.1f} s", flush=True)
