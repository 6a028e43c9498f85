"""bench.py's host-side pieces that run without a GPU: the reference CPU baseline leg (oracle/_ref/ref_harness on the workload's own read generator) and
the profile lookup.  The JSON contract itself needs a device (tests/test_gpu_rccl.py, the driver's run)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_cpu_baseline_leg():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_harness")):
        pytest.skip("the reference binary is built only where /root/reference exists")
    b = _bench()
    r = b.cpu_baseline("chr2M_hifi30x", "full", threads=4)
    assert r["kind"] == "reference" and r["unit"] == "overlaps/s" and r["cores"] == 4 and r["value"] > 0 and r["overlaps"] > 10_000
    assert r["sampled"] is False and "chr2M_hifi30x" in r["sample"]
    r2 = b.cpu_baseline("chr1_250M_hifi30x", "sample", threads=1) if os.environ.get("HAO_TEST_SLOW") else None      # (20 s of reference time: opt-in)
    assert r2 is None or r2["sampled"] is True


def test_profile_lookup_prefers_the_newest_round():
    b = _bench()
    p, rel = b.profile_file("pmc_traffic.json")
    newest = max(d for d in os.listdir(os.path.join(ROOT, "profiles")) if os.path.exists(os.path.join(ROOT, "profiles", d, "pmc_traffic.json")))
    assert rel == f"profiles/{newest}/pmc_traffic.json" and os.path.exists(p)
    assert b.profile_file("no_such_file.json") == (None, None)
    assert {"sketch_unit_kernel", "chain_group_kernel", "seed_lds_kernel", "seed_bin_kernel"} <= set(b.ALG)      # (the seed stage's kernel is named by the engine per run: hao_batch_seed_path)


def test_default_workload_per_gpu_count():
    """no --workload: configs[2] (the largest single-GPU configuration) weak-scaled for 1 - 7 GPUs, configs[3] - the configuration BASELINE.json's metric is quoted on,
    split over the ranks - for 8; with 2 - 7 GPUs configs[3] rides along as variants.metric_config (every rank runs it: the call sits outside `if rank == 0`)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'a.workload = METRIC_WORKLOAD if a.gpus == 8 else "chr1_250M_hifi30x"' in src and 'METRIC_WORKLOAD = "human3G_hifi40x"' in src
    i, j, k = src.index("metric_var = None"), src.index("mv = run_workload(a, METRIC_WORKLOAD"), src.index("    if rank == 0:\n        # SURVEY 8d")
    assert i < j < k
