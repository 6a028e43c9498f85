"""The emulation's answer must not depend on the order in which it runs the lanes of a wave between two rendezvous: by default from the highest lane down (the
common single-writer idiom - all lanes read, then lane 0 or the first lane of a group writes - then behaves as in lockstep by itself), here from the lowest up.
That only works because every place where a kernel relies on lockstep between a read by all lanes and a write by one of them carries HAO_LOCKSTEP() (seven places:
hao_query.cuh, hao_query3.cuh, hao_chain.cuh) or a fence.  All 22 small scenarios pass both ways (HAO_SIMT_ASCENDING=1 python tests/simt_pipeline.py NAME); two run here."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("name", ["hifi", "bf22"])
def test_lowest_lane_first(name):
    r = subprocess.run([sys.executable, os.path.join(HERE, "simt_pipeline.py"), name], capture_output=True, text=True, env=dict(os.environ, HAO_SIMT_ASCENDING="1"))
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout[-400:], r.stderr[-1200:])
