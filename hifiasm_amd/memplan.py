"""Device-memory plan of one rank (launcher-side arithmetic, no GPU): what is resident in each phase of a round - ha_ft_gen, ha_pt_gen, the all-reads pass - for a
workload split over `world` GPUs, from the sizes the allocation sites of hifiasm_amd/csrc use.  bench.py prints it next to a multi-GPU line; tests/test_abi_cpu.py holds
BASELINE.json's configs[3] / configs[4] against the 288 GB of an MI355X with it; tests/test_gpu_zz_rankshare.py measures the ha_ft_gen figure on a device.

Per-unit figures (bytes) and where they come from:
  reads            0.25 / base + 1 / read + 4 / read (lengths of ALL reads, replicated)     hao_set_reads, hao_set_shard
  ft, one pass     20 / k-mer slot (two 8-byte occurrence buffers + 25 %: sort scratch)     hao_tables.hpp HAO_FT_BYTES_PER_SLOT; sharded 46 (+ receive buffer and twin);
                   + 10 through the Bloom filter (two block ids and a flag per occurrence)   HAO_FT_BYTES_PER_SLOT_BLOOM, hao_bloom_filter
  ft, P passes     the above / P + 2 x 8 x 2^28 (the read-chunk scratch)                    hao_ft_run, local_hashes
  ft, run lists    44 / distinct k-mer of the rank's hash range (keys 8 + counts 4, once     hao_sort_rle_hist, hao_keep_runs (flag, position, start: 3 x 8)
                   more while a pass's runs are appended)
  minimizers       24 / local minimizer (hash, record, lookup answer)                       hao_sketch_run, d_ix_lk
  pt partition     60 / minimizer of the rank's hash range while the index is built          DESIGN 6 (sort pairs, arrival index, run ids, scatter windows)
  index            8 / minimizer of ALL reads + 20 / kept key (key, start, count) + 2^26 x 4  hao_pt_run: the all-gather's slots ARE the replicated index
  pass, per batch  130 / seed hit with both delivery sets (k_mer_hit 16, sorted copy 16,     bench.py's batch sizing (measured: 6 batches of 1.07e9 hits of configs[2] = 180 GB with the index)
                   group tables, chain records, fake cigars, codes, two output sets)
Minimizer and seed-hit densities are the REFERENCE's on the full-size fixtures (tests/golden/*.npz: sum of count x histogram of ha_pt_gen; seed hits per read)."""
from __future__ import annotations

HBM_BYTES = 288e9
FT_PER_SLOT, FT_PER_SLOT_SHARDED, FT_RUN_PER_SLOT, FT_CHUNK_SLOTS, FT_PER_SLOT_BLOOM = 20.0, 46.0, 3.0, 1 << 28, 10.0


def ft_passes(slots: float, free_bytes: float, sharded: bool, bloom: bool = False) -> int:
    """hao_ft_pass_count (hao_tables.hpp) for `slots` k-mer slots of local reads and `free_bytes` of free device memory"""
    import math
    per = (FT_PER_SLOT_SHARDED if sharded else FT_PER_SLOT) + (FT_PER_SLOT_BLOOM if bloom else 0.0)
    have = 0.9 * free_bytes
    p_size = max(1, math.ceil(slots / 2 ** 32))      # a pass's occurrence buffers hold at most 2^32 slots each (allocating bigger ones costs more than a second hashing of the reads)
    if per * slots + (1 << 30) <= have:
        return min(64, p_size)
    rest = have - 2 * 8 * FT_CHUNK_SLOTS - (2 << 30) - FT_RUN_PER_SLOT * slots      # (the run lists of all passes: 12 + 12 bytes per distinct k-mer, one per ~7 occurrences allowed for)
    return 64 if rest <= 0 else min(64, max(p_size, math.ceil(per * slots / rest)))


def rank_plan(total_bases: float, n_reads: float, world: int, mz_per_base: float, hits_per_read: float, genome: float, err: float = 0.001, k: int = 51,
              batch_hits: float = 1.07e9, kept_key_frac: float = 0.6, bloom: bool = False) -> dict:
    """bytes resident on one of `world` ranks in each phase; every phase includes what stays from the earlier ones"""
    b_loc, r_loc = total_bases / world, n_reads / world
    reads = 0.25 * b_loc + r_loc + 4 * n_reads
    # distinct k-mers: the genome's (both strands are one canonical k-mer) + those an error makes (a k-mer is error-free with probability (1 - err)^k); a rank counts
    # its hash range: a world-th of them.  Through the Bloom filter (-f37, the reference's default) a k-mer enters the table at its second occurrence: the genome's + 10 %
    novel = total_bases * (1.0 - (1.0 - err) ** k)
    distinct = (1.1 * genome if bloom else genome + novel) / world
    free_ft = HBM_BYTES - reads
    p = ft_passes(b_loc, free_ft, world > 1, bloom)
    per = (FT_PER_SLOT_SHARDED if world > 1 else FT_PER_SLOT) + (FT_PER_SLOT_BLOOM if bloom else 0.0)
    ft_count = reads + per * b_loc / p + (2 * 8 * FT_CHUNK_SLOTS if p > 1 else 0) + 12 * distinct * (1 + (1.0 / p if p > 1 else 0))      # a pass's buffers + the run lists so far
    ft_keep = reads + 12 * distinct + 24 * distinct                                                                                        # the threshold pass over the complete run list
    ft = max(ft_count, ft_keep)
    m_all = mz_per_base * total_bases; m_loc = m_all / world
    keys = kept_key_frac * m_all / 28.0                      # ~ one key per list of ~28 positions (30 - 40 x coverage); a fraction survives the count thresholds
    index = 8 * m_all + 20 * keys + (1 << 26) * 4
    pt = reads + 24 * m_loc + 60 * m_loc + index             # the partition is built while the all-gather's slots exist
    hits_pass = hits_per_read * r_loc
    batch = min(batch_hits, hits_pass)
    query = reads + 24 * m_loc + index + 130 * batch
    return {"passes_ft": p, "reads": reads, "ft_gen": ft, "pt_gen": pt, "all_reads_pass": query, "index": index, "seed_hits_per_pass": hits_pass,
            "batches_per_pass": max(1, round(hits_pass / batch)), "peak": max(ft, pt, query), "hbm": HBM_BYTES}
