"""N > 1 launcher-side logic on CPU: two gloo ranks shard one read set, replicate the read lengths and share
the communicator id exactly as bench.py does before creating the engines (the engines themselves need GPUs)."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from hifiasm_amd import shard, synth


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    genome = synth.make_genome(60_000, seed=5)
    n_total = 37
    lo, hi = shard.shard_range(n_total, rank, world)
    rs = synth.make_reads(genome, hi - lo, 3000, 0.002, seed=6, rid0=lo, len_jit=500)
    all_len, counts = shard.gather_lengths(dist, rs.lengths)
    uid = shard.share_unique_id(dist, lambda: b"U" * 128)
    q.put((rank, lo, hi, rs.lengths.copy(), rs.packed.copy(), all_len, counts, uid))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    genome = synth.make_genome(60_000, seed=5)
    full = synth.make_reads(genome, 37, 3000, 0.002, seed=6, len_jit=500)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == 37           # contiguous ranges in rank order
    for rank, lo, hi, lens, packed, all_len, counts, uid in res:
        assert (lens == full.lengths[lo:hi]).all()                                   # a shard is a slice of the one read set
        assert (packed == full.packed[int(full.pk_off[lo]):int(full.pk_off[hi])]).all()
        assert (all_len == full.lengths).all() and counts == [res[0][2], 37 - res[0][2]]
        assert uid == b"U" * 128


def test_shard_ranges_cover():
    for n in (0, 1, 7, 1000):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


import pytest  # noqa: E402


_RCCL_CASES = [("hifi", 2, {}), ("nn", 3, {}), ("bf24", 2, {}), ("hifi", 4, {}), ("hifi", 2, {"HAO_FT_PASSES": "3", "HAO_DBG_TEST": "ft_chunk_slots=25000"})]


@pytest.mark.parametrize("name,world,env", _RCCL_CASES if os.environ.get("HAO_SIMT_FULL") else [_RCCL_CASES[1], _RCCL_CASES[3], _RCCL_CASES[4]])      # (default suite: 3 ranks with N reads, 4 ranks, 2 ranks in passes; HAO_SIMT_FULL=1: all five)
def test_rccl_branch_between_processes(name, world, env):
    """hao_comm.hpp's RCCL branch with 2, 3 and 4 ranks, one PROCESS per rank under torch.distributed.run: the emulated device library (tests/simt) in every process,
    tests/simt/rccl/rccl.h - grouped send / receive, in-place all-gather, broadcast per root, sum all-reduce over a mailbox directory - in RCCL's place, gloo on the
    launcher side (read lengths, unique id).  Every rank compares its tables and every one of its reads with the oracle (tests/rccl_worker.py; exit code 0 = bit-exact).
    Until a box with two GPUs runs tests/test_gpu_rccl.py this is the only place where the grouped exchange pattern runs with more than one rank."""
    import subprocess
    import sys
    import simt_build
    simt_build.build_lib()      # once, before the ranks start (they would otherwise queue on the build lock)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "tests", "rccl_worker.py"), name, "--simt"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, OMP_NUM_THREADS="1", HAO_SIMT_RCCL_TIMEOUT="600", **env))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("0 differ") == world, r.stdout[-1500:]
