"""Named synthetic workloads (BASELINE.json configs 1-4) shared by bench.py and the full-size parity tests.

Every read has its own counter-based RNG stream (csrc/hao_synth.c), so any rank can generate any contiguous shard of a
workload's read set and the bytes are identical on every machine.
"""
from __future__ import annotations

from . import synth

WORKLOADS = {
    # name: (genome_size, coverage, read_len, err, repeat_rich, is_ont)
    "bacterial5M_hifi30x": (5_000_000, 30, 15000, 0.001, 0, 0),            # BASELINE.json configs[1]
    "bacterial5M_hifi30x_repeat": (5_000_000, 30, 15000, 0.001, 1, 0),     # its repeat-rich variant
    "chr2M_hifi30x": (2_000_000, 30, 15000, 0.001, 0, 0),                  # stand-in for configs[0] (chr11-2M.fa.gz is not in the image)
    "chr1_250M_hifi30x": (250_000_000, 30, 15000, 0.001, 0, 0),            # configs[2]: the largest single-GPU configuration
    "chr1_250M_hifi30x_repeat": (250_000_000, 30, 15000, 0.001, 1, 0),     # configs[2] with the SURVEY 8d repeat recipe (125 units: half of the genome sits in 25-copy families)
    "chr1_250M_hifi30x_jitter": (250_000_000, 30, 16500, 0.001, 0, 0),     # configs[2] with read lengths uniform in 8 - 25 kb (LEN_JIT below): the same bases in reads that are not all alike
    "human3G_hifi40x": (3_000_000_000, 40, 15000, 0.001, 0, 0),            # configs[3]: 8 M reads of 15 kb, sharded over 8 GPUs
    "human375M_hifi40x": (375_000_000, 40, 15000, 0.001, 0, 0),            # a one-GPU proxy of ONE RANK of configs[3] on 8 GPUs: a rank's read count (1 M reads of 15 kb, 15 Gbases) at configs[3]'s coverage,
                                                                           # i.e. its seed-hit density (16 k per read); run with HAO_DBG_TEST=ix_pad=N the index has the replicated index's 3.45 G records
    "ont5M_30x": (5_000_000, 30, 30000, 0.01, 0, 1),
    "ont50M_30x": (50_000_000, 30, 30000, 0.01, 0, 1),                     # 50 000 ONT reads: the full-size parity case of --ont mode
    "ont_human_30x": (3_000_000_000, 30, 30000, 0.01, 0, 1),               # configs[4]: 3 M reads of 30 kb, --ont
}
LEN_JIT = {"chr1_250M_hifi30x_jitter": 8500}      # +- uniform jitter on the read length (workloads not listed: every read exactly read_len bases)
GENOME_SEED, READ_SEED = 11, 12


def n_reads_of(name: str) -> int:
    g, cov, L, _err, _rr, _ont = WORKLOADS[name]
    return max(1, int(round(g * cov / L)))


def workload_reads(name: str, lo: int = 0, hi: int | None = None, want_codes: bool = False, genome=None):
    """reads [lo, hi) of the named workload's read set (default: all of it)"""
    g, _cov, L, err, rr, _ont = WORKLOADS[name]
    if genome is None:
        genome = synth.make_genome(g, seed=GENOME_SEED, repeat_rich=rr)
    hi = n_reads_of(name) if hi is None else hi
    return synth.make_reads(genome, hi - lo, L, err, seed=READ_SEED, rid0=lo, len_jit=LEN_JIT.get(name, 0), want_codes=want_codes)
