#!/usr/bin/env python3
"""bench.py - read-pair overlaps/sec of the MI355X overlap engine (BASELINE.json metric).

One step = ha_pt_gen (sketch all reads + index) + one all-reads h_ec_lchain pass over the
workload, exactly the span BASELINE.md / SURVEY.md 8(d) define:
    overlaps/sec = sum(ol->length) / (t(ha_pt_gen) + t(all-reads pass)).
The read store is resident in HBM before the timed region (hao_set_reads) and ha_ft_gen has
run (it is reported separately).  Default workload = BASELINE.json
configs[1]: synthetic 5 Mb genome, 30x HiFi, 15 kb reads, 0.1 % error, one MI355X.

N > 1 (launched by torch.distributed.run, one rank per GPU): ONE all-vs-all problem whose genome is
N x the single-GPU genome, so every rank owns the same number of query reads (weak scaling).  Reads are
sharded by query read; ha_ft_gen counts k-mers by hash range (RCCL all-to-all-v + 32 KB all-reduce),
ha_pt_gen all-gathers the 16-byte minimizer records so every rank holds the whole index, the query
pass needs no communication (SURVEY.md 8e layout i).  Barrier + max-over-ranks timing,
value = all ranks' overlaps / max time.  If the RCCL communicator cannot be created the ranks fall back
to independent shards and say so in config.parallelism.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (genome_size, coverage, read_len, err, repeat_rich, is_ont)
    "bacterial5M_hifi30x": (5_000_000, 30, 15000, 0.001, 0, 0),
    "bacterial5M_hifi30x_repeat": (5_000_000, 30, 15000, 0.001, 1, 0),
    "chr2M_hifi30x": (2_000_000, 30, 15000, 0.001, 0, 0),
    "chr1_250M_hifi30x": (250_000_000, 30, 15000, 0.001, 0, 0),
    "ont5M_30x": (5_000_000, 30, 30000, 0.01, 0, 1),
}
# algorithmic bytes per unit (stated again in DESIGN.md)
ALG = {
    "sketch_chunk_wave_kernel": ("base", 0.25 + 16.0 / 35.0),       # 2-bit bases in + one 16-B minimizer per ~35 bases out
    "chain_group_kernel": ("anchor", 16 + 4),                        # k_mer_hit in + fake-cigar / record out (hits stay in place)
    "seed_expand_kernel": ("anchor", 8 + 8),                         # index position in + key out
    "chain_assemble_kernel": ("anchor", 16 + 16),                    # chained hit in + tagged hit out
    "seg_bin_sort_kernel": ("anchor", 8 + 16),                       # key in, k_mer_hit out (groups come with the bin table)
    "seed_bin_kernel": ("anchor", 8 + 16),                           # index record in, k_mer_hit out (bins, order and groups in LDS)
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def make_reads(workload, seed, rank=0, world=1):
    """reads [rank*n, (rank+1)*n) of the read set over a genome `world` times the workload's genome"""
    from hifiasm_amd import synth
    g, cov, L, err, rr, ont = WORKLOADS[workload]
    genome = synth.make_genome(g * world, seed=seed, repeat_rich=rr)
    n_reads = max(1, int(round(g * cov / L)))
    return synth.make_reads(genome, n_reads, L, err, seed=seed + 1, rid0=rank * n_reads, want_codes=False), ont


def cpu_baseline(sample_reads=4000, threads=None):
    """The real reference (oracle/_ref/ref_harness, unmodified hifiasm) on a bounded sample of the same
    workload kind, all host cores; falls back to the C restatement (1 thread) when the binary is absent."""
    from hifiasm_amd import synth
    cores = threads or os.cpu_count() or 1
    g, cov, L, err = 2_000_000, 30, 15000, 0.001
    genome = synth.make_genome(g, seed=11)
    rs = synth.make_reads(genome, sample_reads, L, err, seed=12)
    sample = f"{sample_reads} reads x {L} bp, {cov}x of a {g // 1_000_000} Mb i.i.d. genome, 0.1 % error, -f0"
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    if os.path.exists(harness):
        d = tempfile.mkdtemp(prefix="hao_cpu_")
        fa = os.path.join(d, "r.fa")
        synth.write_fasta(fa, rs)
        r = subprocess.run([harness, "-t", str(cores), "--time", fa], capture_output=True, text=True, cwd=d)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            return {"value": j["overlaps_per_sec"], "unit": "overlaps/s", "cores": cores, "kind": "reference", "sample": sample,
                    "t_ft_gen": j["t_ft_gen"], "t_pt_gen": j["t_pt_gen"], "t_pass": j["t_pass"], "overlaps": j["overlaps"]}
        except Exception as ex:  # fall through to the port
            sys.stderr.write(f"[bench] ref_harness failed ({ex}); using the C restatement\n")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    n = min(sample_reads, 600)
    sub_off = rs.code_off[: n + 1]
    o = oracle_py.Oracle(rs.codes[: int(sub_off[-1])], sub_off)
    o.ft_gen()
    t0 = time.time()
    o.pt_gen()
    tot = 0
    for r in range(n):
        tot += o.lchain(r)[0].shape[0]
    dt = time.time() - t0
    return {"value": tot / dt, "unit": "overlaps/s", "cores": 1, "kind": "port", "sample": f"first {n} of: " + sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="bacterial5M_hifi30x", choices=list(WORKLOADS))
    ap.add_argument("--batch-reads", type=int, default=0, help="query reads per hao_overlap_batch (0 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    force_sharded = os.environ.get("HAO_BENCH_FORCE_SHARDED") == "1" and "RANK" in os.environ      # exercise the N > 1 code path with one rank (tests)
    if world > 1 or force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from hifiasm_amd.api import Engine
    from hifiasm_amd import build as _b  # noqa: F401  (libraries are prebuilt; build() is the driver's job)

    mode = "single GPU"
    rs, is_ont = make_reads(a.workload, seed=11, rank=rank, world=world)
    eng = Engine(local_rank, is_ont=is_ont)
    eng.set_readset(rs)
    t0 = time.time()
    hom_ft = None
    if world > 1 or force_sharded:
        ok, why = True, ""
        try:
            # lengths of all reads (replicated, 4 B/read) and the communicator id travel over the launcher's process group
            from hifiasm_amd import shard
            all_len, counts = shard.gather_lengths(dist, rs.lengths, device="cuda")
            uid = shard.share_unique_id(dist, Engine.dist_unique_id)
            eng.set_shard(sum(counts[:rank]), all_len)
            eng.dist_init(uid, rank, world)
            hom_ft = eng.ha_ft_gen()
            eng.ha_pt_gen()               # probe: every collective of the sharded build has run once before anything is timed
            mode = f"reads sharded by query over {world} GPUs; RCCL: k-mer all-to-all-v by hash range, minimizer all-gather (replicated index), no query-time traffic"
        except Exception as ex:  # noqa: BLE001
            ok, why = False, repr(ex)
        # all ranks take the same path: one failure anywhere sends everybody to independent shards
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            sys.stderr.write(f"[bench] rank {rank}: sharded mode unavailable ({why or 'another rank failed'}); running independent shards\n")
            eng.close()
            rs, is_ont = make_reads(a.workload, seed=11 + 1000 * rank)
            eng = Engine(local_rank, is_ont=is_ont)
            eng.set_readset(rs)
            hom_ft = None
            mode = f"FALLBACK: {world} independent shards (own genome per rank), no data-path collective"
    if hom_ft is None:
        hom_ft = eng.ha_ft_gen()
    t_ft = time.time() - t0
    n_reads = rs.n
    # hao_overlap_batch handles < 2^32 seed hits per call: ~12.4 k hits per 15 kb read at 30x -> cap the batch
    # and ~100 B of device scratch per seed hit: keep a batch near 4e8 hits (~40 GB)
    auto_bsz = max(1, int(4e8 // max(1.0, 0.83 * rs.total_bases / max(1, n_reads))))
    bsz = a.batch_reads if a.batch_reads > 0 else min(n_reads, auto_bsz)

    def step():
        eng.ha_pt_gen()
        st = {k: v for k, v in eng.stage_times()}
        tot = {"overlaps": 0, "chained_hits": 0, "seed_hits": 0, "groups": 0, "minimizers": 0, "seq_groups": 0, "seq_group_hits": 0}
        for lo in range(0, n_reads, bsz):
            eng.overlap_batch(lo, min(n_reads, lo + bsz))
            t = eng.batch_totals()
            for k in tot:
                tot[k] += t[k]
            for k, v in eng.stage_times():
                st[k] = st.get(k, 0.0) + v
        return tot, st

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.time()
    stage_sum = {}
    for _ in range(a.steps):
        tot, st = step()
        for k, v in st.items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v
    sync()
    dt = time.time() - t0
    overlaps = tot["overlaps"]
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        oo = torch.tensor([overlaps], dtype=torch.int64, device="cuda")
        dist.all_reduce(oo, op=dist.ReduceOp.SUM)
        overlaps = int(oo.item())
    ms_per_step = dt / a.steps * 1e3
    value = overlaps / (dt / a.steps)

    out = None
    if rank == 0:
        stage_ms = {k: v / a.steps for k, v in stage_sum.items()}
        # roofline of the dominant kernel: live HIP-event time of its stage on the engine's stream
        kern_stage = {"sketch_chunk_wave_kernel": "sk_chunks", "chain_group_kernel": "q_chain", "seed_expand_kernel": "q_expand",
                      "chain_assemble_kernel": "q_assemble", "seed_bin_kernel": "q_sort_bins"}
        dom = max(kern_stage, key=lambda k: stage_ms.get(kern_stage[k], 0.0))
        unit, bpu = ALG[dom]
        units = rs.total_bases if unit == "base" else tot["seed_hits"]
        n_batches = 1 if unit == "base" else max(1, (n_reads + bsz - 1) // bsz)
        # per batch: all launches of the kernel in one batch count as one "launch" (chain_group_kernel runs once per size class,
        # seed_bin_kernel once per bin-table size); the stage time brackets exactly those launches
        k_ms = stage_ms.get(kern_stage[dom], 0.0) / n_batches
        alg_bytes = bpu * units / n_batches
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))
            if pj.get("workload") == a.workload and world == 1:
                tt = [v["hbm_bytes_per_launch"] * v["launches"] for k, v in pj["kernels"].items() if k.split("<")[0] == dom]
                traffic = int(sum(tt) / max(1, pj.get("batches", 1))) if tt else None
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel_ms": round(k_ms, 4), "alg_bytes_per_launch": int(alg_bytes)}
        out = {
            "metric": "read-pair overlaps/sec (sum ol->length / (ha_pt_gen + all-reads h_ec_lchain pass))",
            "value": round(value, 1), "unit": "overlaps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 hashing / int32 + f64 chain scores", "data": "synthetic",
            "config": {"workload": a.workload, "reads_per_gpu": n_reads, "bases_per_gpu": rs.total_bases,
                       "overlaps_per_gpu_step": tot["overlaps"], "seed_hits_per_gpu_step": tot["seed_hits"],
                       "chained_hits_per_gpu_step": tot["chained_hits"], "groups_per_gpu_step": tot["groups"],
                       "groups_on_sequential_path": tot["seq_groups"], "k": 51, "w": 51, "hpc": 1,
                       "parallelism": mode,
                       "ha_ft_gen_s": round(t_ft, 3), "hom_cov_ft": hom_ft},
            "roofline": roofline,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        elif not a.no_cpu_baseline:
            out["cpu_baseline"] = None
        if a.verbose:
            sys.stderr.write(json.dumps(out["stage_ms"], indent=1) + "\n")
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
