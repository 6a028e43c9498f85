"""The RCCL transport with more than one rank (one process per GPU, launched like the driver launches bench.py).  Needs >= 2 GPUs: skipped on a
1-GPU box, where tests/test_gpu_sharded.py covers the same code above the transport through the loopback backend."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("name", ["hifi", "rr", "bf24"])
@pytest.mark.parametrize("world", [2, 4])
def test_rccl_world(name, world):
    if _ngpu() < world:
        pytest.skip(f"{_ngpu()} GPU(s) visible, {world} needed")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + world), os.path.join(ROOT, "tests", "rccl_worker.py"), name]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher) must start two RCCL ranks and report n_gpus = 2"""
    import json
    if _ngpu() < 2:
        pytest.skip("needs 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "chr2M_hifi30x",
                        "--cpu-baseline", "none", "--no-boundary"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and "RCCL" in j["config"]["parallelism"]


def test_bench_refuses_more_gpus_than_visible():
    n = _ngpu()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "visible" in r.stderr
