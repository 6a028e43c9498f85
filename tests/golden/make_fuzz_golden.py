"""TEST INFRASTRUCTURE: per-read digests of the oracle's results for the random workloads of tests/simt_fuzz.py whose oracle run is too long for the GPU suite (repeat-dense
genomes: groups of thousands of seed hits through the chain DP - minutes on a CPU core).  `python tests/golden/make_fuzz_golden.py OUT.npz SEED` writes one seed's arrays
(coverage peaks, per read: crc32 of the sketch, digest of the seed hits, digest of (ol, fake cigars, chained hits) - helpers.digest_hits / digest_result);
`python tests/golden/make_fuzz_golden.py --merge DIR OUT.npz` folds the per-seed files into tests/golden/fuzz_heavy.npz.  tests/test_gpu_fuzz.py compares libhao.so's
results with them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(seed):
    import simt_fuzz
    import oracle_py
    from helpers import crc, digest_hits, digest_result
    rs, d, okw = simt_fuzz.reads_of(seed)
    o = oracle_py.Oracle(rs.codes, rs.code_off, **okw)
    ft = o.ft_gen(); hom = o.pt_gen(); het = o.stats()["het_cov"]
    sk = np.zeros(rs.n, dtype=np.uint32); hd = np.zeros(rs.n, dtype=np.uint64); rd = np.zeros(rs.n, dtype=np.uint64); tot = 0
    for r in range(rs.n):
        sk[r] = crc(o.sketch(r)); hd[r] = digest_hits(o.seed_hits(r))
        ol, fc, fo, cl = o.lchain(r); rd[r] = digest_result(ol, fc, cl); tot += ol.shape[0]
    return dict(peaks=np.array([ft, hom, het, rs.n, tot], dtype=np.int64), sketch=sk, hits=hd, result=rd)


if __name__ == "__main__":
    if sys.argv[1] == "--merge":
        out = {}
        for f in sorted(os.listdir(sys.argv[2])):
            if f.endswith(".npz"):
                z = np.load(os.path.join(sys.argv[2], f))
                for k in z.files:
                    out[f"s{f[:-4]}_{k}"] = z[k]
        np.savez_compressed(sys.argv[3], **out)
        print(len(out) // 4, "seeds ->", sys.argv[3])
    else:
        np.savez(sys.argv[1], **one(int(sys.argv[2])))
