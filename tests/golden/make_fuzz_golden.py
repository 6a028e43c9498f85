"""TEST INFRASTRUCTURE: tests/golden/fuzz_heavy.npz - per-read digests of the REAL reference's results (oracle/_ref/ref_harness --digest: the unmodified hifiasm
sources compiled from /root/reference) for the random workloads of tests/simt_fuzz.py whose restatement run is too long for the GPU suite (repeat-dense genomes:
groups of thousands of seed hits through the chain DP).  Per seed: the coverage peaks, and per read the crc32 of its minimizers, the digest of its seed hits and the
digest of (ol, fake cigars, chained hits) - the definitions of helpers.digest_hits / digest_result = hao_batch_digest (include/hao.h).  tests/test_gpu_fuzz.py
compares libhao.so's results with them; tests/test_fuzz_ref_cpu.py recomputes a few.  (Round 5's file came from the C restatement; the reference gives the same
arrays for all 68 seeds.)

    python tests/golden/make_fuzz_golden.py [WORKERS]        (build container only; the slowest case takes five minutes of the reference)"""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(seed):
    import ref_fuzz
    return seed, ref_fuzz.digests(seed, threads=2)


if __name__ == "__main__":
    import test_gpu_fuzz
    out = {}
    with ProcessPoolExecutor(int(sys.argv[1]) if len(sys.argv) > 1 else 6) as ex:
        for seed, g in ex.map(one, test_gpu_fuzz.HEAVY):
            for k, v in g.items():
                out[f"s{seed}_{k}"] = v
            print("seed", seed, "reads", int(g["peaks"][3]), "overlaps", int(g["peaks"][4]), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "fuzz_heavy.npz")
    np.savez_compressed(path, **out)
    print(len(out) // 4, "seeds ->", path)
