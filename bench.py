#!/usr/bin/env python3
"""bench.py - read-pair overlaps/sec of the MI355X overlap engine (BASELINE.json metric).

One step = ha_pt_gen (sketch all reads + index) + one all-reads h_ec_lchain pass over the workload, exactly the span
BASELINE.md / SURVEY.md 8(d) define:
    overlaps/sec = sum(ol->length) / (t(ha_pt_gen) + t(all-reads pass)).
The read store is resident in HBM before the timed region (hao_set_reads) and ha_ft_gen has run (reported separately, as in
BASELINE.md 2b).  Default workload = BASELINE.json configs[2], the largest single-GPU configuration: synthetic 250 Mb genome,
30x HiFi, 500 000 reads of 15 kb, 0.1 % error (`--workload bacterial5M_hifi30x` = configs[1]).

`value` is the rate of the step with every batch's results (ol->list, fake cigars, cl->list in the packed wire format of
include/hao.h) delivered into pinned host memory through the streaming path (hao_overlap_batch_async: the download of batch i runs
under the compute of batch i+1) - what h_ec_lchain's callers actually get; `value_resident` is the same step with the results left in
HBM (the per-kernel roofline figures come from that run: every kernel alone on the device).  `--no-boundary` measures only the latter.

`--gpus N` with N > 1 launches N ranks itself (torch.distributed.run, one rank per GPU, RCCL) when it was not started by a launcher;
under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  N > 1: ONE all-vs-all problem whose genome is N x the single-GPU genome, so
every rank owns the same number of query reads (weak scaling; `human3G_hifi40x` and `ont_human_30x` are fixed-size problems split over
the ranks: strong scaling).  Reads are sharded by query read; ha_ft_gen counts k-mers by hash range (RCCL all-to-all-v + 32 KB
all-reduce), ha_pt_gen builds the index hash-partitioned and all-gathers it (SURVEY.md 8e layout i: replicated index, no query-time
traffic).  Barrier + max-over-ranks timing, value = all ranks' overlaps / max time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hifiasm_amd.workloads import WORKLOADS, LEN_JIT, n_reads_of  # noqa: E402

STRONG = {"human3G_hifi40x", "ont_human_30x"}      # fixed-size problems: the read set is split over the ranks
VARIANT_OF = {"chr1_250M_hifi30x": "chr1_250M_hifi30x_repeat", "bacterial5M_hifi30x": "bacterial5M_hifi30x_repeat"}      # the repeat-rich twin of a workload
# a one-GPU PROXY of one rank of configs[3] on 8 GPUs (no multi-GPU box has been available in any round): a rank's read count at configs[3]'s coverage - i.e. its
# seed-hit density - over an index padded (HAO_DBG_TEST=ix_pad=N) to the replicated index's 3.45 G position records.  What it does NOT contain: the exchanges of
# ha_ft_gen / ha_pt_gen (all-to-all-v, one 27.6 GB all-gather per round) and target ids spread over 8 M reads.  8 x its rate minus that all-gather is a PREDICTION of the
# 8-GPU metric, not a measurement: no scaling curve has been measured.
RANK_PROXY_OF = {"chr1_250M_hifi30x": ("human375M_hifi40x", 3_450_000_000)}
JITTER_OF = {"chr1_250M_hifi30x": "chr1_250M_hifi30x_jitter"}      # the same bases in reads of 8 - 25 kb (uniform): the timed numbers of the headline workload see reads that are all exactly 15 000 bases
# algorithmic bytes per unit (stated again in DESIGN.md 4)
ALG = {
    "sketch_unit_kernel": ("base", 0.25 + 16.0 / 35.0),             # 2-bit bases in + one 16-B minimizer per ~35 bases out
    "chain_group_kernel": ("anchor", 16 + 4),                        # k_mer_hit in + fake-cigar / record out (hits stay in place)
    "seed_lds_kernel": ("anchor", 8 + 16),                           # index record in (once, coalesced), k_mer_hit out (the read's position lists staged in LDS and merged by target there, hao_query5.cuh)
    "seed_bin_kernel": ("anchor", 8 + 16),                           # (repeat-rich batches; HAO_SEED_LDS=0: the table kernels of rounds 1 - 4) index record in, k_mer_hit out
}
METRIC_WORKLOAD = "human3G_hifi40x"      # BASELINE.json configs[3]: the configuration the metric is quoted on (8 GPUs)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2      # G wave-level VALU instructions / s: 256 CUs x 4 SIMD-32 x 2.4 GHz, a wave64 instruction issues over 2 cycles (MI355X_MICROARCH.md)


def make_reads(workload, rank=0, world=1):
    """this rank's shard of the workload's read set (weak scaling: genome x world; strong: the fixed set split by rank)"""
    from hifiasm_amd import synth
    from hifiasm_amd.workloads import GENOME_SEED, READ_SEED
    g, cov, L, err, rr, ont = WORKLOADS[workload]
    if workload in STRONG:
        genome = synth.make_genome(g, seed=GENOME_SEED, repeat_rich=rr)
        n_all = n_reads_of(workload)
        lo, hi = n_all * rank // world, n_all * (rank + 1) // world
    else:
        genome = synth.make_genome(g * world, seed=GENOME_SEED, repeat_rich=rr)
        n = n_reads_of(workload)
        lo, hi = rank * n, (rank + 1) * n
    return synth.make_reads(genome, hi - lo, L, err, seed=READ_SEED, rid0=lo, len_jit=LEN_JIT.get(workload, 0), want_codes=False), ont


def cpu_baseline(workload, mode="sample", threads=None):
    """The real reference (oracle/_ref/ref_harness = unmodified hifiasm objects) on the SAME workload: the benched read set itself
    (mode "full") or, by default, a bounded sample of it - the same generator and parameters over the first <= 50 Mb of genome (100 000
    reads of 15 kb: ~10-30 s of work on all host cores, enough to keep them busy).  Falls back to the C restatement (1 thread) when the
    reference binary is absent."""
    from hifiasm_amd import synth
    from hifiasm_amd.workloads import GENOME_SEED, READ_SEED
    cores = threads or os.cpu_count() or 1
    g, cov, L, err, rr, ont = WORKLOADS[workload]
    gs = g if mode == "full" else min(g, 50_000_000)
    n = max(1, int(round(gs * cov / L)))
    genome = synth.make_genome(gs, seed=GENOME_SEED, repeat_rich=rr)
    sample = (f"{workload}: the benched read set itself" if gs == g else
              f"{workload} generator over a {gs // 1_000_000} Mb genome") + f" ({n} reads x {L} bp, {cov}x, {err * 100:g} % error, -f0)"
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):
        base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 3 * gs * cov else None
        d = tempfile.mkdtemp(prefix="hao_cpu_", dir=base)
        try:
            fa = os.path.join(d, "r.fq" if ont else "r.fa")
            synth.write_fasta_stream(fa, genome, n, L, err, seed=READ_SEED, len_jit=LEN_JIT.get(workload, 0), fastq=bool(ont))
            cmd = [harness, "-t", str(cores), "--time"] + (["--ont"] if ont else []) + [fa]
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
            j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            return {"value": j["overlaps_per_sec"], "unit": "overlaps/s", "cores": cores, "kind": "reference", "sample": sample, "sampled": gs != g,
                    "t_ft_gen": j["t_ft_gen"], "t_pt_gen": j["t_pt_gen"], "t_pass": j["t_pass"], "overlaps": j["overlaps"]}
        except Exception as ex:  # fall through to the port
            sys.stderr.write(f"[bench] ref_harness failed ({ex}); using the C restatement\n")
        finally:
            shutil.rmtree(d, ignore_errors=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    nn = min(n, 600)
    rs = synth.make_reads(genome, nn, L, err, seed=READ_SEED)
    o = oracle_py.Oracle(rs.codes, rs.code_off, is_ont=int(ont))
    o.ft_gen()
    t0 = time.time()
    o.pt_gen()
    tot = 0
    for r in range(nn):
        tot += o.lchain(r)[0].shape[0]
    dt = time.time() - t0
    return {"value": tot / dt, "unit": "overlaps/s", "cores": 1, "kind": "port", "sample": f"first {nn} reads of: " + sample}


def profile_file(name):
    """newest committed profile of that name (PMC counters need their own rocprofv3 passes - tools/r06_final.sh pmc - so they cannot be
    collected inside this run; the line says where the figure comes from)"""
    for r in ("r06", "r05", "r04", "r03", "r02"):
        p = os.path.join(ROOT, "profiles", r, name)
        if os.path.exists(p):
            return p, f"profiles/{r}/{name}"
    return None, None


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run and relay rank 0's line"""
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        sys.stderr.write(f"[bench] --gpus {a.gpus} but only {have} GPU(s) are visible\n")
        sys.exit(2)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def verify_delivery(eng, workload, rs, dranges):
    """one delivered pass, untimed: per-read digests of what landed in host memory, folded over blocks of 256 reads, against the reference's"""
    import numpy as np
    fx = os.path.join(ROOT, "tests", "golden", workload + ".npz")
    res = {"reads": rs.n, "fixture": os.path.relpath(fx, ROOT) if os.path.exists(fx) else None, "equal_to_reference": None}
    eng.ha_pt_gen()
    dig = np.zeros(rs.n, dtype=np.uint64); n_cl = 0
    prev = None
    for lo, hi in list(dranges) + [(None, None)]:
        slot = eng.overlap_batch_async(lo, hi) if lo is not None else None
        if prev is not None:
            d = eng.deliver_wait(prev[0])
            dig[prev[1]:prev[2]] = eng.delivery_digest(d); n_cl += int(d.n_cl)
        prev = (slot, lo, hi) if lo is not None else None
    res["chained_hits_decoded"] = n_cl
    # fold over blocks of 256 reads: sum of mix64(d_r + GOLD * (r + 1)) - the definition tests/golden/make_golden_big.py folds the reference's digests with
    with np.errstate(over="ignore"):
        z_ = dig + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, dig.size + 1, dtype=np.uint64)
        z_ ^= z_ >> np.uint64(30); z_ *= np.uint64(0xbf58476d1ce4e5b9); z_ ^= z_ >> np.uint64(27); z_ *= np.uint64(0x94d049bb133111eb); z_ ^= z_ >> np.uint64(31)
        f = np.zeros((dig.size + 255) // 256, dtype=np.uint64)
        np.add.at(f, np.arange(dig.size) // 256, z_)
    res["fold_crc"] = int(__import__("zlib").crc32(f.tobytes()))
    if res["fixture"]:
        z = np.load(fx)
        res["equal_to_reference"] = bool(z["dig_fold"].shape == f.shape and (z["dig_fold"] == f).all())
        if not res["equal_to_reference"]:
            sys.stderr.write(f"[bench] DELIVERED RESULTS DIFFER FROM THE REFERENCE on {workload}\n")
    return res


def run_workload(a, workload, steps, warmup, rank, local_rank, world, dist, torch, force_sharded, check_delivery=True):
    """one workload on this rank's GPU: resident and boundary-inclusive timings, roofline of the dominant kernel; returns the line's dict on rank 0 (None elsewhere)"""
    from hifiasm_amd.api import Engine
    mode = "single GPU"
    rs, is_ont = make_reads(workload, rank=rank, world=world)
    eng = Engine(local_rank, is_ont=is_ont)
    eng.set_readset(rs)
    t0 = time.time()
    if world > 1 or force_sharded:
        # lengths of all reads (replicated, 4 B/read) and the communicator id travel over the launcher's process group.  No fallback: the engine's
        # collectives carry every rank's status (hao_comm.hpp), so a failure raises on ALL ranks instead of leaving some of them inside a collective.
        from hifiasm_amd import shard
        all_len, counts = shard.gather_lengths(dist, rs.lengths, device="cuda")
        uid = shard.share_unique_id(dist, Engine.dist_unique_id)
        eng.set_shard(sum(counts[:rank]), all_len)
        eng.dist_init(uid, rank, world)
        mode = (f"reads sharded by query over {world} GPUs (RCCL over xGMI): k-mer all-to-all-v by hash range, hash-partitioned index build + "
                f"all-gather (SURVEY 8e layout i: replicated index, no query-time traffic; layout ii - seed hits routed to the target's owner - not built)")
    hom_ft = eng.ha_ft_gen()
    t_ft = time.time() - t0
    n_reads = rs.n
    # hao_overlap_batch handles < 2^32 seed hits per call and needs ~130 B of device scratch per seed hit: a batch of ~8e8 hits (~110 GB with the index;
    # 62 500 reads of configs[2]).  Bigger batches amortise the tails of the per-batch kernels and of the DP side streams (configs[2]: 16 batches 226 ms,
    # 8 batches 214 ms, 4 batches 209 ms per step at 200 GB)
    # (a repeat-rich genome: ~1.4 x the seed hits per read, and more of them in exception lists / DP scratch)
    # (round 4: ~1.07e9 hits per batch on the repeat-free sets = 6 batches of configs[2], 180 of the device's 309 GB with both delivery sets - 174.7 instead of 179.3 ms resident,
    # and two exposed copy tails less; the repeat-rich twin keeps 12 batches: 198 GB, nine would need 240)
    auto_bsz = max(1, int(8e8 // max(1.0, (1.25 if WORKLOADS[workload][4] else 0.62) * rs.total_bases / max(1, n_reads) * WORKLOADS[workload][1] / 30.0)))
    bsz = a.batch_reads if a.batch_reads > 0 else min(n_reads, auto_bsz)
    ranges = [(lo, min(n_reads, lo + bsz)) for lo in range(0, n_reads, bsz)]
    # with delivery a pass needs at least two batches for the copy of one to run under the compute of the next: a pass that fits one batch is cut in two
    # (more pieces hide more of the copy but pay the tails of the per-batch kernels once per piece: four pieces measured slower on every small workload)
    dranges = ranges if len(ranges) >= 2 else [(n_reads * i // 2, n_reads * (i + 1) // 2) for i in range(2) if n_reads * (i + 1) // 2 > n_reads * i // 2]
    # The copy of a pass's LAST batch has no kernels to run under: ~0.87 GB = 15 ms on configs[2].  Rounds 4 - 5 cut the last range into 1/2 + 1/4 + 1/4 (--tail-split keeps
    # the A/B): once a batch's copy ends before the next batch's kernels do, that cut makes the small pieces wait for the big one's copy - 183.7 ms with it, 180.7 without.
    # What works is a TAPER (--taper, default 0.8,0.55,0.4,0.25; "none": equal batches): the last two batches re-cut into pieces that shrink step by step, so that every
    # piece's copy still ends under the next, smaller piece's kernels and only a quarter batch's copy is left exposed - same box: 181.7 / 181.1 ms with equal batches,
    # 178.0 / 177.0 with three pieces (0.85,0.7,0.45), 175.8 with four.
    if len(dranges) >= 4 and a.tail_split:
        lo_, hi_ = dranges[-1]; m1, m2 = lo_ + (hi_ - lo_) // 2, lo_ + 3 * (hi_ - lo_) // 4
        if lo_ < m1 < m2 < hi_:
            dranges = dranges[:-1] + [(lo_, m1), (m1, m2), (m2, hi_)]
    # (the consumer takes batch k between the kernels of k + 1 and k + 2: a piece's copy has to end before the NEXT piece's kernels do, or the device waits for the host)
    if len(dranges) >= 4 and a.taper and a.taper != "none" and not a.tail_split:
        fr = [float(x) for x in a.taper.split(",")]
        lo_, hi_ = dranges[-2][0], dranges[-1][1]
        if abs(sum(fr) - 2.0) < 1e-6 and all(f > 0 for f in fr):
            cuts = [lo_]; acc = 0.0
            for f in fr[:-1]:
                acc += f; cuts.append(lo_ + int(round((hi_ - lo_) * acc / 2.0)))
            cuts.append(hi_)
            if all(x < y for x, y in zip(cuts[:-1], cuts[1:])):
                dranges = dranges[:-2] + list(zip(cuts[:-1], cuts[1:]))

    views = {}

    def contexts(k):      # the engine plus k - 1 attached batch contexts (own stream, scratch and result buffers over the same index)
        while len(views) < k - 1:
            views[len(views)] = eng.attach()
        return [eng] + [views[i] for i in range(k - 1)]

    def step(deliver=False, n_ctx=1):
        eng.ha_pt_gen()
        st = {k: v for k, v in eng.stage_times()}
        keys0 = {"overlaps": 0, "chained_hits": 0, "seed_hits": 0, "groups": 0, "minimizers": 0, "seq_groups": 0, "seq_group_hits": 0, "host_bytes": 0, "copy_ms": 0.0, "t_async": 0.0, "t_wait": 0.0, "code_bytes": 0, "delivered_hits": 0, "exceptions": 0}
        engs = contexts(n_ctx)
        res, errs = [None] * n_ctx, []

        def run(ci):      # context ci takes batches ci, ci + n_ctx, ...
            e = engs[ci]; tot = dict(keys0); sst = {}
            try:
                prev = None
                for lo, hi in (dranges if deliver else ranges)[ci::n_ctx]:
                    if deliver:
                        t_a = time.time()
                        slot = e.overlap_batch_async(lo, hi)      # compute of this batch; its copy runs under the next batch's kernels
                        t_b = time.time()
                        if prev is not None:                      # the consumer takes the previous batch now (its copy ran under this batch's kernels)
                            d = e.deliver_wait(prev)
                            tot["host_bytes"] += int(d.bytes); tot["copy_ms"] += float(d.copy_ms); tot["code_bytes"] += int(d.n_codes); tot["delivered_hits"] += int(d.n_cl); tot["exceptions"] += int(d.n_exc)
                        tot["t_async"] += (t_b - t_a) * 1e3; tot["t_wait"] += (time.time() - t_b) * 1e3
                        prev = slot
                    else:
                        e.overlap_batch(lo, hi)
                    t = e.batch_totals()
                    for k in t:
                        if k in tot:
                            tot[k] += t[k]
                    for k, v in e.stage_times():
                        sst[k] = sst.get(k, 0.0) + v
                if deliver and prev is not None:
                    d = e.deliver_wait(prev)                      # every batch's results are in host memory
                    tot["host_bytes"] += int(d.bytes); tot["copy_ms"] += float(d.copy_ms); tot["code_bytes"] += int(d.n_codes); tot["delivered_hits"] += int(d.n_cl); tot["exceptions"] += int(d.n_exc)
                res[ci] = (tot, sst)
            except Exception as ex:      # noqa: BLE001 - raised again on the main thread
                errs.append(ex)

        if n_ctx == 1:
            run(0)
        else:
            import threading
            th = [threading.Thread(target=run, args=(ci,)) for ci in range(n_ctx)]
            [t.start() for t in th]; [t.join() for t in th]
        if errs:
            raise errs[0]
        tot = dict(keys0)
        for t_, s_ in res:
            for k in tot:
                tot[k] += t_[k]
            for k, v in s_.items():      # (with several contexts the stage times are sums over concurrently running batches: they no longer add up to the step)
                st[k] = st.get(k, 0.0) + v
        return tot, st

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def timed(deliver, n_ctx=1):
        for _ in range(warmup):
            step(deliver, n_ctx)
        sync()
        t0 = time.time()
        ssum = {}
        for _ in range(steps):
            tot, st = step(deliver, n_ctx)
            for k, v in st.items():
                ssum[k] = ssum.get(k, 0.0) + v
        sync()
        dt = time.time() - t0
        ov = tot["overlaps"]
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            oo = torch.tensor([ov], dtype=torch.int64, device="cuda")
            dist.all_reduce(oo, op=dist.ReduceOp.SUM)
            ov = int(oo.item())
        return dt, ov, tot, ssum

    dt, overlaps, tot, stage_sum = timed(False, max(1, a.contexts))
    ms_per_step = dt / steps * 1e3
    value = overlaps / (dt / steps)
    boundary = None
    if not a.no_boundary and hasattr(eng, "overlap_batch_async"):
        bdt, bov, btot, bst = timed(True, max(1, a.boundary_contexts))
        boundary = {"q_assemble_ms_per_step": bst.get("q_assemble", 0.0) / steps, "stage_ms": {k: round(v / steps, 2) for k, v in bst.items()},"value": bov / (bdt / steps), "ms_per_step": bdt / steps * 1e3, "host_bytes_per_gpu_step": btot["host_bytes"], "wire_bytes_per_chained_hit": (btot["delivered_hits"] / 8 + btot["delivered_hits"] / 16 + btot["code_bytes"]) / max(1, btot["delivered_hits"]),
                    "copy_ms_per_step": btot["copy_ms"], "verbatim_hits_per_step": btot["exceptions"], "host_ms_in_async": btot["t_async"], "host_ms_in_wait": btot["t_wait"], "copy_gb_per_s": btot["host_bytes"] / max(1e-9, btot["copy_ms"] * 1e-3) / 1e9}

    # What `value` was quoted on must be what the reference computes: one more delivered pass (untimed), every read's digest computed on the HOST from the bytes
    # in the pinned arena (hao_delivery_digest: ol->list, fake cigars, cl->list decoded out of the wire format) against the digests of the real reference's run on
    # the same reads (tests/golden/<workload>.npz, made by tests/golden/make_golden_big.py from oracle/_ref/ref_harness; compared by value, nothing under oracle/ runs here)
    if boundary is not None and world == 1 and not a.no_verify:
        boundary["delivered_bytes_check"] = verify_delivery(eng, workload, rs, dranges)

    if rank == 0:
        stage_ms = {k: v / steps for k, v in stage_sum.items()}
        # roofline of the dominant kernel: live HIP-event time of its stage on the engine's stream
        # which kernels carried the seed stage: the engine says (hao_batch_seed_path: the choice is made per batch in hao_batch.hpp - repeat-rich batches keep the table kernels)
        try:
            seed_path = eng.batch_seed_path()
        except Exception:      # (the owner context ran no batch of its own: --contexts > 1)
            seed_path = {"first_launch": "seed_bin_kernel" if os.environ.get("HAO_SEED_LDS") == "0" else "seed_lds_kernel", "left_to_tables": None}
        seed_kernel = seed_path["first_launch"]
        KERN_STAGE = {"sketch_unit_kernel": "sk_chunks", "chain_group_kernel": "q_chain", seed_kernel: "q_sort_bins"}
        dom = max(KERN_STAGE, key=lambda k: stage_ms.get(KERN_STAGE[k], 0.0))
        unit, bpu = ALG[dom]
        units = rs.total_bases if unit == "base" else tot["seed_hits"]
        n_batches = 1 if unit == "base" else len(ranges)
        # per batch: all launches of the kernel in one batch count as one "launch" (chain_group_kernel runs once per size class,
        # seed_bin_kernel once per bin-table size); the stage time brackets exactly those launches
        k_ms = stage_ms.get(KERN_STAGE[dom], 0.0) / n_batches
        alg_bytes = bpu * units / n_batches
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "kernel_ms": round(k_ms, 4), "alg_bytes_per_launch": int(alg_bytes)}
        # the same figure for each of the three kernels that carry the step (the dominant one is `roofline` itself)
        def kernel_line(kn):
            u_, b_ = ALG[kn]; un_ = rs.total_bases if u_ == "base" else tot["seed_hits"]; nb_ = 1 if u_ == "base" else len(ranges)
            ms_ = stage_ms.get(KERN_STAGE[kn], 0.0) / nb_
            ach_ = b_ * un_ / nb_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0.0
            return {"kernel": kn, "kernel_ms": round(ms_, 4), "launches_per_step": nb_, "alg_bytes_per_launch": int(b_ * un_ / nb_), "achieved": round(ach_, 2), "frac": round(ach_ / HBM_PEAK_GBS, 5)}
        roofline["kernels"] = [kernel_line(kn) for kn in KERN_STAGE]
        # what the device gives this byte mix (8 in + 16 out per seed hit) with no computation attached: profiles/r05/ubench_gather.txt (tools/ubench_gather.hip)
        roofline["seed_stage"] = {"first_launch": seed_kernel, "reads_left_to_the_table_kernels_in_the_last_batch": seed_path["left_to_tables"]}
        # what the device gives this byte mix (8 in + 16 out per seed hit) with no computation attached: constants measured by tools/ubench_gather.hip, NOT in this run
        ub, ub_rel = profile_file("ubench_gather.json")
        if ub:
            try:
                roofline["seed_stage_ceilings"] = dict(json.load(open(ub)), source=ub_rel + " (tools/ubench_gather.hip, its own run)")
            except Exception:
                pass
        prof, prof_rel = profile_file("pmc_traffic.json")
        if prof and world == 1:      # PMC counters need their own rocprofv3 passes (tools/r06_final.sh pmc): not measurable inside this run
            try:
                pj = json.load(open(prof))
                if pj.get("workload") == workload:
                    tt = [v["hbm_bytes_per_launch"] * v["launches"] for k, v in pj["kernels"].items() if k.split("<")[0] == dom]
                    if tt:
                        roofline["traffic"] = int(sum(tt) / max(1, n_batches))      # (the profile's launches of one pass, spread over this run's launches per pass)
                        roofline["traffic_source"] = prof_rel + ": separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH_SIZE x 2 (gfx950 correction of MI355X_MICROARCH.md; calibrated for wide coalesced reads only - for this kernel's 8-byte gathers the true value lies between traffic_lo and traffic)"
                        lo_ = [v.get("hbm_bytes_per_launch_raw", v["hbm_bytes_per_launch"]) * v["launches"] for k, v in pj["kernels"].items() if k.split("<")[0] == dom]
                        roofline["traffic_lo"] = int(sum(lo_) / max(1, n_batches))
            except Exception:
                pass
        # the sketch kernel is instruction-issue bound, not HBM bound: report its VALU issue rate next to the HBM fraction (counters: profiles/r0N/sketch_alu.json)
        sk_ms = stage_ms.get("sk_chunks", 0.0)
        sk = {"kernel": "sketch_unit_kernel", "kernel_ms": round(sk_ms, 4), "bases": rs.total_bases,
              "gbases_per_s": round(rs.total_bases / (sk_ms * 1e-3) / 1e9, 2) if sk_ms > 0 else None,
              "hbm_frac": round(ALG["sketch_unit_kernel"][1] * rs.total_bases / (sk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if sk_ms > 0 else None}
        alu, alu_rel = profile_file("sketch_alu.json")
        if alu:
            try:
                aj = json.load(open(alu))
                if not str(aj.get("kernel", "")).startswith("sketch_unit_kernel"):
                    raise KeyError("counters of another kernel")
                per_base = aj["valu_wave_insts_per_base"]
                ach = per_base * rs.total_bases / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else 0.0
                sk["roofline_alu"] = {"achieved_valu_issue": round(ach, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                                      "frac": round(ach / VALU_PEAK_GINST, 4), "valu_wave_insts_per_base": per_base,
                                      "source": alu_rel + " (rocprofv3 --pmc SQ_INSTS_VALU ...) x this run's kernel time",
                                      "peak_is": "the guide's 2 cycles per wave64 instruction; measured on the device (profiles/r06/ubench_valu.txt) only the simple two-operand "
                                                 "integer instructions issue in 2.2 - 2.6 cycles, everything else in 4.0 - 4.5: this kernel's mix runs at its issue limit (DESIGN 8)"}
            except Exception:
                pass
        out = {
            "metric": "read-pair overlaps/sec (sum ol->length / (ha_pt_gen + all-reads h_ec_lchain pass))",
            "value": round(boundary["value"] if boundary else value, 1), "unit": "overlaps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(boundary["ms_per_step"] if boundary else ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if workload in STRONG else "weak",
            "vs_baseline": None, "dtype": "u64 hashing / int32 + f64 chain scores", "data": "synthetic",
            "value_is": ("boundary: every batch's ol->list, fake cigars and cl->list (wire format) delivered into pinned host memory inside the timed region" if boundary
                         else "resident: results stay in HBM (--no-boundary)"),
            "value_resident": round(value, 1), "ms_per_step_resident": round(ms_per_step, 3),
            "value_boundary": round(boundary["value"], 1) if boundary else None,
            "boundary": ({"ms_per_step": round(boundary["ms_per_step"], 3), "host_bytes_per_gpu_step": boundary["host_bytes_per_gpu_step"], "wire_bytes_per_chained_hit": round(boundary["wire_bytes_per_chained_hit"], 4),
                          "copy_ms_per_step": round(boundary["copy_ms_per_step"], 2), "verbatim_hits_per_step": boundary["verbatim_hits_per_step"], "copy_gb_per_s": round(boundary["copy_gb_per_s"], 2),
                          "assemble_and_pack_ms_per_step": round(boundary["q_assemble_ms_per_step"], 2),
                          "host_ms_in_async": round(boundary["host_ms_in_async"], 1), "host_ms_in_wait": round(boundary["host_ms_in_wait"], 1),
                          "stage_ms": boundary["stage_ms"], "delivered_bytes_check": boundary.get("delivered_bytes_check"),
                          "contexts": max(1, a.boundary_contexts), "batches_per_pass": len(dranges), "batch_reads": [hi_ - lo_ for lo_, hi_ in dranges] if len(dranges) <= 16 else None,
                          "what": "same step with every batch's ol->list, fake cigars and packed cl->list delivered into pinned host memory (per batch context: double-buffered, copy stream under the next batch's compute); contexts = batch contexts (hao_attach), one host thread each, that share the pass"}
                         if boundary else None),
            "config": {"workload": workload, "batch_contexts": max(1, a.contexts), "reads_per_gpu": n_reads, "bases_per_gpu": rs.total_bases, "batches_per_pass": len(ranges),
                       "overlaps_per_gpu_step": tot["overlaps"], "seed_hits_per_gpu_step": tot["seed_hits"],
                       "chained_hits_per_gpu_step": tot["chained_hits"], "groups_per_gpu_step": tot["groups"],
                       "groups_on_sequential_path": tot["seq_groups"], "k": 51, "w": 51, "hpc": 1,
                       "parallelism": mode, "ha_ft_gen_s": round(t_ft, 3), "hom_cov_ft": hom_ft},
            "roofline": roofline,
            "device_memory": (lambda fr_to: {"used_gb_at_end_of_run": round((fr_to[1] - fr_to[0]) / 1e9, 1), "total_gb": round(fr_to[1] / 1e9, 1)})(torch.cuda.mem_get_info(local_rank)),
            "sketch": sk,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        }
        for v_ in views.values():
            v_.close()
        eng.close()
        return out
    for v_ in views.values():
        v_.close()
    eng.close()
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS), help="default: chr1_250M_hifi30x (BASELINE configs[2], weak-scaled by --gpus); with --gpus 8: human3G_hifi40x (configs[3], the configuration the metric is quoted on, split over the ranks)")
    ap.add_argument("--batch-reads", type=int, default=0, help="query reads per hao_overlap_batch (0 = sized for ~8e8 seed hits)")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full", "none"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary-inclusive (results delivered to host memory) measurement")
    ap.add_argument("--contexts", type=int, default=1, help="batch contexts (hao_attach) = host threads that run the batches of a pass concurrently; 1 keeps every kernel alone on the device (the roofline line)")
    ap.add_argument("--boundary-contexts", type=int, default=1, help="batch contexts of the boundary-inclusive measurement")
    ap.add_argument("--no-rank-proxy", action="store_true", help="skip variants.rank_proxy_configs3 (1 M reads at 40x over an index padded to configs[3]'s size: a minute of generation and ha_ft_gen)")
    ap.add_argument("--no-variants", action="store_true", help="skip the repeat-rich twin of the workload (the `variants` block of the line)")
    ap.add_argument("--variant-steps", type=int, default=5, help="timed steps of the variant (at most --steps)")
    ap.add_argument("--taper", default="0.8,0.55,0.4,0.25", help="delivered pass: sizes (in batches, sum 2) the last two batches are re-cut into; none: equal batches")
    ap.add_argument("--tail-split", action="store_true", help="delivered pass: cut the last batch into 1/2 + 1/4 + 1/4 (A/B: the default of rounds 4 - 5)")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed pass that digests the delivered bytes and compares them with the reference's digests")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    if a.no_cpu_baseline:
        a.cpu_baseline = "none"
    if a.workload is None:
        a.workload = METRIC_WORKLOAD if a.gpus == 8 else "chr1_250M_hifi30x"

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and not (world == 1 and os.environ.get("HAO_BENCH_FORCE_SHARDED") == "1"):
        sys.stderr.write(f"[bench] --gpus {a.gpus} disagrees with WORLD_SIZE={world}\n")
        sys.exit(2)
    import torch
    dist = None
    force_sharded = os.environ.get("HAO_BENCH_FORCE_SHARDED") == "1" and "RANK" in os.environ      # exercise the N > 1 code path with one rank (tests)
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count() <= local_rank:
            sys.stderr.write(f"[bench] rank {rank}: no GPU {local_rank} ({torch.cuda.device_count()} visible)\n")
            sys.exit(2)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from hifiasm_amd.api import Engine

    out = run_workload(a, a.workload, a.steps, a.warmup, rank, local_rank, world, dist, torch, force_sharded)
    # 2 <= N < 8 ranks: the line's headline stays the weak-scaled configs[2] (comparable across N); the configuration the metric is quoted on - configs[3], the
    # fixed 8 M-read set split over the N ranks - rides along as variants.metric_config (one untimed + one timed step, results left in HBM) when the memory plan of
    # a rank says it fits (hifiasm_amd/memplan.py); every rank takes part
    metric_var = None
    if world > 1 and not a.no_variants and a.workload != METRIC_WORKLOAD:
        from hifiasm_amd import memplan
        g_, cov_, L_, err_ = WORKLOADS[METRIC_WORKLOAD][:4]
        plan = memplan.rank_plan(float(g_) * cov_, n_reads_of(METRIC_WORKLOAD), world, 0.02873, 0.92 * 0.02873 * L_ * cov_, float(g_), err=err_)
        if plan["peak"] < 0.9 * plan["hbm"]:
            nb_ = a.no_boundary; a.no_boundary = True
            mv = run_workload(a, METRIC_WORKLOAD, 1, 1, rank, local_rank, world, dist, torch, force_sharded)
            a.no_boundary = nb_
            if rank == 0:
                metric_var = {k: mv[k] for k in ("value", "ms_per_step", "value_resident", "ms_per_step_resident", "steps", "warmup", "scaling", "roofline", "stage_ms", "config")}
                metric_var["memory_plan_gb"] = {k: round(v / 1e9, 1) for k, v in plan.items() if isinstance(v, float)}
        elif rank == 0:
            metric_var = {"skipped": f"a rank's share of {METRIC_WORKLOAD} on {world} GPUs does not fit: plan peak {plan['peak'] / 1e9:.0f} GB"}
    if rank == 0:
        # SURVEY 8d / BASELINE.md 2b: "report both variants" - the same step on the repeat-rich read set of the same size (filter table, minimizer thinning,
        # max_n_chain pruning, the chain DP: the case that looks like a real genome), fewer steps
        out["variants"] = None
        var = VARIANT_OF.get(a.workload)
        if var and world == 1 and not a.no_variants:
            v = run_workload(a, var, max(1, min(a.steps, a.variant_steps)), min(a.warmup, 1), rank, local_rank, world, dist, torch, force_sharded)
            out["variants"] = {"repeat_rich": {k: v[k] for k in ("value", "ms_per_step", "value_resident", "ms_per_step_resident", "steps", "warmup", "roofline", "stage_ms", "config", "boundary")}}
        jit = JITTER_OF.get(a.workload)
        if jit and world == 1 and not a.no_variants:
            v = run_workload(a, jit, max(1, min(a.steps, a.variant_steps)), min(a.warmup, 1), rank, local_rank, world, dist, torch, force_sharded)
            out["variants"] = dict(out["variants"] or {}, length_jitter={k: v[k] for k in ("value", "ms_per_step", "value_resident", "ms_per_step_resident", "steps", "warmup", "roofline", "stage_ms", "config", "boundary")})
        prox = RANK_PROXY_OF.get(a.workload)
        if prox and world == 1 and not a.no_variants and not a.no_rank_proxy:
            from hifiasm_amd import memplan
            pw, ix_records = prox
            g_, cov_, L_, err_ = WORKLOADS[pw][:4]
            n_mz_ = int(0.02873 * g_ * cov_)      # (minimizers of the proxy's own reads: the pad brings the index to the replicated index's size)
            os.environ["HAO_DBG_TEST"] = "ix_pad=" + str(max(0, ix_records - n_mz_))
            try:
                v = run_workload(a, pw, 1, 1, rank, local_rank, world, dist, torch, force_sharded)
            finally:
                del os.environ["HAO_DBG_TEST"]
            g3_, cov3_, L3_, err3_ = WORKLOADS[METRIC_WORKLOAD][:4]
            plan = memplan.rank_plan(float(g3_) * cov3_, n_reads_of(METRIC_WORKLOAD), 8, 0.02873, 0.92 * 0.02873 * L3_ * cov3_, float(g3_), err=err3_)
            pv = {k: v[k] for k in ("value", "ms_per_step", "value_resident", "ms_per_step_resident", "steps", "warmup", "roofline", "stage_ms", "config", "boundary", "device_memory")}
            pv["index_records_with_pad"] = ix_records
            pv["memory_plan_gb_of_a_configs3_rank"] = {k: (round(x / 1e9, 1) if isinstance(x, float) else x) for k, x in plan.items()}
            allgather_s = 8.0 * ix_records * 7 / 8 / (7 * 153e9 * 0.7)      # a rank receives 7/8 of the 27.6 GB index over its 7 xGMI links at ~70 % of 153 GB/s each
            pv["prediction_8_gpus"] = {"overlaps_per_s": round(8 * v["config"]["overlaps_per_gpu_step"] / (v["ms_per_step"] * 1e-3 + allgather_s), 1),
                                       "assumed_allgather_s": round(allgather_s, 4),
                                       "what": "8 x this rank-sized pass + one all-gather of the 27.6 GB index per round; a PREDICTION - no scaling curve was measured, RCCL with >= 2 ranks has not run"}
            out["variants"] = dict(out["variants"] or {}, rank_proxy_configs3=pv)
        if metric_var is not None:
            out["variants"] = dict(out["variants"] or {}, metric_config=metric_var)
        if world > 1:      # what one rank holds, phase by phase (launcher-side arithmetic from the allocation sites' sizes: hifiasm_amd/memplan.py)
            from hifiasm_amd import memplan
            g_, cov_, L_, err_ = WORKLOADS[a.workload][:4]
            tb_ = float(g_) * cov_ * (1 if a.workload in STRONG else world)
            pl_ = memplan.rank_plan(tb_, tb_ / L_, world, 0.02873, 0.92 * 0.02873 * L_ * cov_, float(g_) * (1 if a.workload in STRONG else world), err=err_, bloom=err_ > 0.005)
            out["memory_plan_gb"] = {k: (round(v / 1e9, 1) if isinstance(v, float) else v) for k, v in pl_.items()}
        out["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_baseline) if (a.cpu_baseline != "none" and world == 1) else None
        if a.verbose:
            sys.stderr.write(json.dumps(out["stage_ms"], indent=1) + "\n")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
