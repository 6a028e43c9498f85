"""A one-GPU PROXY of one rank of BASELINE.json configs[3] (3 Gb, 40x, 8 M reads on 8 GPUs), against the REAL reference: `human375M_hifi40x` = 1 M reads of 15 kb
over a 375 Mb genome - a rank's read count at configs[3]'s coverage, i.e. its seed-hit density (16 k per read) - with HAO_DBG_TEST=ix_pad=N bringing the position index to the
replicated index's 3.45 G records (27.6 GB): ha_pt_gen and the all-reads pass at a rank's size, which the rank-share test of ha_ft_gen (test_gpu_zz_rankshare.py)
left open.  The fixture (tests/golden/human375M_hifi40x.npz) is the unmodified reference's run on the same reads (tests/golden/make_golden_big.py): coverage peaks,
thresholds, minimizer histogram, totals, and a digest of EVERY read's seed hits and (ol, fake cigars, cl).

What the proxy does not contain: the exchanges of a sharded round (all-to-all-v of hashes and minimizers, one all-gather of the index) and target ids spread over 8 M
reads.  No multi-GPU box was available in any round: no scaling curve has been measured; bench.py prints the same run as variants.rank_proxy_configs3 with a
PREDICTION of the 8-GPU rate next to it."""
import os
import time
import zlib

import numpy as np
import pytest

from helpers import load_golden, fold_digests, GOLDEN, device_mem_info

pytestmark = pytest.mark.gpu
NAME = "human375M_hifi40x"
IX_RECORDS = 3_450_000_000      # minimizers of configs[3]'s 8 M reads = position records of the replicated index


def test_a_configs3_rank_sized_round_against_the_reference():
    if not os.path.exists(os.path.join(GOLDEN, NAME + ".npz")):
        pytest.skip(f"no fixture tests/golden/{NAME}.npz (tests/golden/make_golden_big.py)")
    from hifiasm_amd import memplan
    from hifiasm_amd.api import Engine
    from hifiasm_amd.workloads import WORKLOADS, workload_reads, n_reads_of
    g = load_golden(NAME); m = g["meta"]
    rs = workload_reads(NAME)
    assert zlib.crc32(rs.lengths.tobytes()) == int(g["len_crc"][0]) and zlib.crc32(rs.packed[: 1 << 20].tobytes()) == int(g["len_crc"][1]), "the synthetic read generator drifted"
    os.environ["HAO_DBG_TEST"] = "ix_pad=" + str(IX_RECORDS - 431_000_000)
    try:
        e = Engine(0)
    finally:
        del os.environ["HAO_DBG_TEST"]
    try:
        e.set_readset(rs)
        t0 = time.time(); hom_ft = e.ha_ft_gen(); t_ft = time.time() - t0
        t0 = time.time(); hom, het = e.ha_pt_gen(); t_pt = time.time() - t0
        assert hom_ft == m["hom_cov_ft"] and (hom, het) == (m["hom_cov"], m["het_cov"])
        st = e.stats()
        assert (st["high_occ"], st["low_occ"], st["max_n_chain"]) == (m["high_occ"], m["low_occ"], m["max_n_chain"])
        assert (e.hist(1) == g["pt_hist"]).all()
        batch = 64_000
        dig = np.zeros(rs.n, dtype=np.uint64); dkh = np.zeros(rs.n, dtype=np.uint64); tot_ol = tot_cl = tot_kh = 0; left = 0
        t0 = time.time()      # (every call below returns with its results on the host)
        for lo in range(0, rs.n, batch):
            hi = min(rs.n, lo + batch)
            e.overlap_batch(lo, hi)
            t = e.batch_totals(); tot_ol += t["overlaps"]; tot_cl += t["chained_hits"]; tot_kh += t["seed_hits"]
            sp = e.batch_seed_path(); assert sp["first_launch"] == "seed_lds_kernel", sp      # 40x repeat-free reads: 16 k hits per read, one per (minimizer x coverage)
            left += sp["left_to_tables"]
            d, k = e.batch_digest(hi - lo); dig[lo:hi] = d; dkh[lo:hi] = k
        t_pass = time.time() - t0
        fr, to = device_mem_info()
        assert tot_ol == m["pass_overlaps"] and tot_cl == m["pass_chained_hits"]
        f, fk = fold_digests(dig), fold_digests(dkh)
        assert (fk == g["dig_kh_fold"]).all(), f"seed hits differ in read blocks {np.flatnonzero(fk != g['dig_kh_fold'])[:10]}"
        assert (f == g["dig_fold"]).all(), f"results differ in read blocks {np.flatnonzero(f != g['dig_fold'])[:10]}"
        g3, cov3, L3, err3 = WORKLOADS["human3G_hifi40x"][:4]
        plan = memplan.rank_plan(float(g3) * cov3, n_reads_of("human3G_hifi40x"), 8, 0.02873, 0.92 * 0.02873 * L3 * cov3, float(g3), err=err3)
        print(f"[rank proxy] {rs.n} reads, {tot_kh} seed hits ({tot_kh / rs.n:.0f} per read), {tot_ol} overlaps: ha_ft_gen {t_ft:.1f} s ({e.ft_passes()} passes), ha_pt_gen {t_pt * 1e3:.0f} ms, "
              f"pass {t_pass * 1e3:.0f} ms in {-(-rs.n // batch)} batches (digests included), {left} reads left to the table kernels; device memory in use at the end {(to - fr) / 1e9:.0f} GB "
              f"(plan of a configs[3] rank: index {plan['index'] / 1e9:.1f} GB, all-reads pass {plan['all_reads_pass'] / 1e9:.0f} GB, {plan['batches_per_pass']} batches)")
    finally:
        e.close()
