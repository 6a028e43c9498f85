"""One rank of the RCCL parity run (launched by tests/test_gpu_rccl.py under torch.distributed.run, one rank per GPU): shard a scenario's reads by
rank, build the tables through the RCCL transport (k-mer all-to-all-v, index all-to-all-v + all-gather, histogram all-reduce), run the query
pass and compare EVERY local read with the oracle.  Exit code 0 = this rank's results are bit-exact.

`--simt` (tests/test_dist_cpu.py, no GPU): the same run with the emulated device library (tests/simt) in every process and tests/simt/rccl/rccl.h - a mailbox
transport between processes - in RCCL's place; the launcher side uses gloo.  What executes is hao_comm.hpp's RCCL branch with two and four ranks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from hifiasm_amd.api import Engine
    from hifiasm_amd import shard
    from hifiasm_amd.synth import ReadSet
    from helpers import scenario_reads, scenario_oracle
    name = sys.argv[1]
    simt = "--simt" in sys.argv[2:]
    rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    if simt:
        import simt_build
        from hifiasm_amd import api
        path = simt_build.build_lib(); api.lib_path = lambda: path; api._LIB = None
        dist.init_process_group("gloo", rank=rank, world_size=world)
        lr = 0
    else:
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
    rs, okw = scenario_reads(name)
    lo, hi = shard.shard_range(rs.n, rank, world)
    sub = ReadSet(lo, rs.lengths[lo:hi].copy(), rs.packed[int(rs.pk_off[lo]):int(rs.pk_off[hi])].copy(), (rs.pk_off[lo:hi + 1] - rs.pk_off[lo]).copy(),
                  rs.codes[int(rs.code_off[lo]):int(rs.code_off[hi])].copy(), (rs.code_off[lo:hi + 1] - rs.code_off[lo]).copy())
    e = Engine(lr, **okw)
    e.set_readset(sub)
    all_len, counts = shard.gather_lengths(dist, sub.lengths, device="cpu" if simt else "cuda")
    assert (all_len == rs.lengths).all()
    e.set_shard(sum(counts[:rank]), all_len)
    e.dist_init(shard.share_unique_id(dist, Engine.dist_unique_id), rank, world)
    e.ha_ft_gen()
    e.ha_pt_gen()
    o = scenario_oracle(name)
    assert e.stats() == o.stats(), (e.stats(), o.stats())
    assert (e.hist(0) == o.ft_hist()).all() and (e.hist(1) == o.pt_hist()).all()
    k, off, pos = e.pt_table(); ok, ooff, opos = o.pt_table()
    assert (k == ok).all() and (off == ooff).all() and (pos == opos).all()
    e.overlap_batch(0, hi - lo)
    bad = 0
    for r in range(lo, hi):
        ol, fc, fo, cl = e.h_ec_lchain(r - lo)
        ool, ofc, ofo, ocl = o.lchain(r)
        bad += int(not (ol.shape == ool.shape and (ol == ool).all() and (fc == ofc).all() and cl.shape == ocl.shape and (cl == ocl).all()))
    e.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"[rccl_worker] rank {rank}/{world} {name}: {hi - lo} reads, {bad} differ", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
