"""The emulation's answer must not depend on the schedule it happens to use.
  * Lanes of a wave between two rendezvous: by default from the highest lane down (the common single-writer idiom - all lanes read, then lane 0 or the first lane
    of a group writes - then behaves as in lockstep by itself); HAO_SIMT_ASCENDING=1 runs them from the lowest up.  That works because every place where a kernel
    relies on lockstep between a read by all lanes and a write by one of them carries HAO_LOCKSTEP() (seven places: hao_query.cuh, hao_query3.cuh, hao_chain.cuh)
    or a fence.
  * Waves of a workgroup: by default wave 0 runs until it stands at a barrier, then wave 1, ...; HAO_SIMT_WAVES=reverse takes them from the last down,
    HAO_SIMT_FAIR=1 advances every wave by one rendezvous in turn.  All are legal interleavings on the device: a kernel whose result changed would be missing a barrier.
All 22 small scenarios pass with the lanes lowest-first, eight of them with the two other wave schedules (`HAO_SIMT_...=... python tests/simt_pipeline.py NAME`); a few run here."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name,env", [("hifi", {"HAO_SIMT_ASCENDING": "1"}), ("bf22", {"HAO_SIMT_ASCENDING": "1"}), ("hifi", {"HAO_SIMT_WAVES": "reverse"}), ("ont", {"HAO_SIMT_FAIR": "1"})])
def test_other_schedules(name, env):
    r = subprocess.run([sys.executable, os.path.join(HERE, "simt_pipeline.py"), name], capture_output=True, text=True, env=dict(os.environ, **env))
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout[-400:], r.stderr[-1200:])


@pytest.mark.parametrize("name", ["hifi", "bf22"])
def test_big_index_path_on_a_small_read_set(name):
    """ha_pt_gen's path for >= 2^23 minimizers (the index sort on 40 hash bits + the fix-up of the runs that hold several keys, hao_index_gather_kernel, one radix
    pass, the windowed scatter of the lookup results) - on the device only the full-size fixtures reach it; HAO_DBG_TEST=sort40_min=N lowers the threshold for the emulation"""
    r = subprocess.run([sys.executable, os.path.join(HERE, "simt_pipeline.py"), name], capture_output=True, text=True, env=dict(os.environ, HAO_DBG_TEST="sort40_min=1", HAO_SIMT_PROF="1"))
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout[-400:], r.stderr[-1200:])
    if os.environ.get("HAO_SIMT_PROF"):      # (tools/simt_coverage.py collects these lines)
        sys.__stderr__.write("\n".join(l for l in r.stderr.splitlines() if l.startswith("[simt prof]")) + "\n")
    assert "hao_index_gather_kernel" in r.stderr and "hao_sort40_mark_kernel" in r.stderr and "hao_scatter_u64_kernel" in r.stderr
