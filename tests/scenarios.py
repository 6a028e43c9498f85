"""Small seeded read sets shared by the golden-fixture generator and the tests.

Each scenario is (synth.dataset kwargs, oracle option overrides).  They are chosen to
walk every branch of the reference hot path: plain HiFi, repeat-rich (non-empty
high-count table, second-level minimizer thinning, real chain DP, multi-copy chains,
max_n_chain pruning), ONT-like error, N bases, too-low coverage (peak_hom = -1: every
k-mer lands in the filter table) and an even k (strand-symmetric k-mer skip) with a
narrower window.
"""
SCENARIOS = {
    "hifi":  (dict(genome_size=40_000, coverage=18, read_len=4000, err=0.002, seed=3, len_jit=1000), {}),
    "rr":    (dict(genome_size=100_000, coverage=12, read_len=4000, err=0.001, seed=5, repeat_rich=2, len_jit=1000), {}),
    "ont":   (dict(genome_size=50_000, coverage=20, read_len=6000, err=0.01, seed=6, len_jit=2000), dict(is_ont=1)),
    "nn":    (dict(genome_size=40_000, coverage=18, read_len=4000, err=0.002, seed=7, len_jit=1000, n_rate=0.0008), {}),
    "low":   (dict(genome_size=40_000, coverage=3, read_len=4000, err=0.002, seed=8), {}),
    "k40":   (dict(genome_size=40_000, coverage=18, read_len=4000, err=0.002, seed=10), dict(k=40, w=30)),
    # Bloom filter in front of the k-mer counts (-f; hifiasm's default is -f37): 2^24 bits = 8 blocks per sub-table here, so false
    # positives (over-counted k-mers, singletons that enter the table) are frequent; repeat-rich so that the filter table is not empty
    "bf24":  (dict(genome_size=100_000, coverage=12, read_len=4000, err=0.001, seed=12, repeat_rich=2, len_jit=1000), dict(bf_shift=24)),
    # 2 blocks per sub-table: the filter saturates, most k-mers are counted one too many and the peaks move
    "bf22":  (dict(genome_size=60_000, coverage=14, read_len=4000, err=0.002, seed=13, repeat_rich=1, len_jit=1000), dict(bf_shift=22)),
    # homopolymer compression off (HA_F_NO_HPC): k-mer spans = k, positions are plain base offsets
    "hpc0":  (dict(genome_size=40_000, coverage=18, read_len=4000, err=0.002, seed=14, len_jit=1000), dict(hpc=0)),
    # the reference's DEFAULT filter size -f37: 2^16 blocks per sub-table, 28-bit block ids; repeat-rich so the table is not empty
    "f37":   (dict(genome_size=100_000, coverage=12, read_len=4000, err=0.001, seed=15, repeat_rich=2, len_jit=1000), dict(bf_shift=37)),
    # the final-round call site (ecovlp.cpp:3957): bw_thres = 0.001 - on 1 % error reads most chain extensions exceed the band
    "bw001": (dict(genome_size=50_000, coverage=20, read_len=6000, err=0.01, seed=16, len_jit=2000), dict(bw_thres=0.001)),
    "bw001rr": (dict(genome_size=100_000, coverage=12, read_len=4000, err=0.001, seed=17, repeat_rich=2, len_jit=1000), dict(bw_thres=0.001)),
    # --hg-size: prior homozygous coverage = total bases / hg_size (htab.cpp:1156,1254; adj_m_peak_hom hist.cpp:46-72).  hg = the genome
    # size: the prior sits on the peak; hg2 = half of it: the only peak lies far below the prior and becomes the heterozygous peak
    "hg":    (dict(genome_size=40_000, coverage=18, read_len=4000, err=0.002, seed=3, len_jit=1000), dict(hg_size=40_000)),
    "hg2":   (dict(genome_size=40_000, coverage=18, read_len=4000, err=0.002, seed=3, len_jit=1000), dict(hg_size=20_000)),
    # corrected-read-like error rate (the final round runs on corrected reads): most overlaps are exact matches (exact-overlap check, f2)
    "exact": (dict(genome_size=60_000, coverage=16, read_len=4000, err=0.0002, seed=18, len_jit=1500, n_rate=0.0001), dict(bw_thres=0.001)),
    # ragged / degenerate reads mixed into a normal set (see edge_reads below)
    "edge":  (dict(builder="edge"), {}),
    # off-default (k, w) pairs and option mixes, chosen to leave the tuned paths: small k with a narrow window, the largest k (63) with a window wider than
    # the wave kernel's chunk overlap and no HPC, ONT mode with N bases, a tiny Bloom filter on a repeat-rich set, error-free reads, and k = 57 (the widest
    # k the one-word window kernel takes) at the default w
    "fz0":   (dict(genome_size=30_000, coverage=14, read_len=2500, err=0.004, seed=40, len_jit=800), dict(k=21, w=11)),
    "fz1":   (dict(genome_size=40_000, coverage=16, read_len=5000, err=0.001, seed=41, len_jit=1500, repeat_rich=1), dict(k=63, w=80, hpc=0)),
    "fz2":   (dict(genome_size=30_000, coverage=18, read_len=3500, err=0.012, seed=42, len_jit=1200, n_rate=0.0005), dict(k=31, w=19, is_ont=1)),
    "fz3":   (dict(genome_size=60_000, coverage=10, read_len=3000, err=0.002, seed=43, len_jit=900, repeat_rich=2), dict(bf_shift=20)),
    "fz4":   (dict(genome_size=25_000, coverage=25, read_len=1800, err=0.0, seed=44, len_jit=600), dict(k=45, w=25)),
    "fz5":   (dict(genome_size=35_000, coverage=15, read_len=4500, err=0.003, seed=45, len_jit=1000), dict(k=57, w=51)),
}


def edge_reads():
    """150 ordinary 3 kb reads plus: reads of 1, 2, 10, k-1, k, k+1, w+k-3 .. w+k-1 and 150 bases, two exact copies of a read, a 4 kb
    homopolymer (one HPC base), a dinucleotide repeat (every minimizer identical), an all-N read, a read with an N every 97 bases
    (no k-mer survives), a read with a single N, and the reverse complement of a read."""
    import numpy as np
    from hifiasm_amd import synth
    g = synth.make_genome(30_000, seed=77)
    base = synth.make_reads(g, 150, 3000, 0.002, seed=78, len_jit=800)
    reads = [base.codes[int(base.code_off[i]):int(base.code_off[i + 1])].copy() for i in range(base.n)]
    rng = np.random.default_rng(5)
    extra = []
    for L in (1, 2, 10, 50, 51, 52, 100, 101, 102, 150):
        st = int(rng.integers(0, 20000))
        extra.append(g[st:st + L].copy())
    extra.append(reads[3].copy())
    extra.append(reads[3].copy())
    extra.append(np.zeros(4000, dtype=np.uint8))
    extra.append(np.tile(np.array([0, 1], dtype=np.uint8), 2000))
    extra.append(np.full(300, 4, dtype=np.uint8))
    x = reads[7].copy(); x[::97] = 4; extra.append(x)
    x = reads[9].copy(); x[1500] = 4; extra.append(x)
    x = reads[11][::-1].copy(); extra.append((3 - x).astype(np.uint8))
    out = []
    for i, r in enumerate(reads):
        out.append(r)
        if i % 8 == 0 and extra:
            out.append(extra.pop(0))
    out += extra
    return synth.from_codes(out)


def len65535_reads():
    """long reads cut so that the longest has EXACTLY 65 535 bases: the last set whose positions fit 16 bits (the seed kernel's 6-byte LDS records, the 4-byte minimizer
    tables of the wire format); one base more and the engine switches to the 32-bit forms ("long200k")"""
    from hifiasm_amd import synth
    rs = synth.dataset(genome_size=400_000, coverage=10, read_len=64_000, err=0.006, seed=43, len_jit=3000)
    reads = [rs.codes[int(rs.code_off[i]):int(rs.code_off[i + 1])][:65535].copy() for i in range(rs.n)]
    assert max(len(r) for r in reads) == 65535 and sum(len(r) == 65535 for r in reads) >= 2
    return synth.from_codes(reads)


def build_reads(dkw):
    """the read set of a scenario: synth.dataset(**dkw), or a hand-made set"""
    from hifiasm_amd import synth
    if dkw.get("builder") == "edge":
        return edge_reads()
    if dkw.get("builder") == "len65535":
        return len65535_reads()
    return synth.dataset(**dkw)


# larger sets used only by the GPU parity tests (oracle vs HIP, no golden file): enough repeat content to push
# thousands of groups through the chain DP / multi-copy / max_n_chain code, and 15 kb reads through the chunked sketch
BIG_SCENARIOS = {
    "rr_big":   (dict(genome_size=400_000, coverage=30, read_len=8000, err=0.001, seed=5, repeat_rich=1, len_jit=2000), {}),
    "hifi_15k": (dict(genome_size=300_000, coverage=25, read_len=15000, err=0.001, seed=21, len_jit=3000), {}),
    "ont_big":  (dict(genome_size=200_000, coverage=30, read_len=12000, err=0.01, seed=6, len_jit=4000), dict(is_ont=1)),
    # stress the capacity fallbacks: > 4096 minimizers per read (query table read from global memory), > 2048-hit groups
    # (chain DP arrays in global scratch), and - in rr_heavy, a genome that is mostly overlapping repeat copies - reads with
    # more than 1024 chains (selection keys in global scratch, bitonic finish)
    "long200k": (dict(genome_size=1_000_000, coverage=16, read_len=200_000, err=0.008, seed=31, len_jit=20_000), dict(is_ont=1)),
    "rr_heavy": (dict(genome_size=300_000, coverage=20, read_len=6000, err=0.001, seed=9, repeat_rich=1, len_jit=1500), {}),
    "len65535": (dict(builder="len65535"), dict(is_ont=1)),                  # 1800 - 1930 minimizers per read: the table kernels
    "len65535w": (dict(builder="len65535"), dict(is_ont=1, w=101)),         # the same reads at w = 101: ~950 minimizers per read, i.e. the list-major seed kernel with 16-bit positions up to 65 534
    # long reads over a repeat-rich genome: groups of several thousand hits that fail the quick check (chain DP with f/p/marks in global scratch)
    "long_rr":  (dict(genome_size=600_000, coverage=10, read_len=100_000, err=0.004, seed=33, repeat_rich=1, len_jit=20_000), dict(is_ont=1)),
}
