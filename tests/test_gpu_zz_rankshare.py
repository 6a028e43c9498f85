"""BASELINE.json configs[3]'s share of ONE of eight ranks through ha_ft_gen on one device: reads [0, 1 000 000) of the 8 M reads of the 3 Gb / 40x HiFi set - 15 Gbases,
11 G HPC k-mer occurrences.  ha_ft_gen chooses its hash-range passes by itself (hao_ft_pass_count, hao_tables.hpp; htab.cpp:707-882 never holds all occurrences
either: 4096 sub-tables filled batch by batch, :147-151, 594-606): on its own a device of 309 GB has the memory for the two 8-byte buffers of 11 G occurrences at once,
but a pass holds fewer than 2^32 occurrences (32-bit slot numbers): three passes (the choice must be what hifiasm_amd/memplan.py predicts from the free memory and that
limit); as a rank of eight - with the receive buffer and its twin beside them - it needs more (forced here: six); both must stay inside the device and give the same
histogram and filter table, the six-pass run in no more memory.  The measured peaks are printed next to the plan's figures."""
import json
import os
import threading
import time

import numpy as np
import pytest

from helpers import device_mem_info

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _PeakPoll(threading.Thread):
    """smallest free device memory seen while it runs (hipMemGetInfo of the runtime libhao.so runs on - helpers.device_mem_info: a driver query, 5 ms apart)"""

    def __init__(self):
        super().__init__(daemon=True)
        from hifiasm_amd import api
        api.lib()      # (the runtime must be in the process before it is asked)
        self.stop = False
        self.free0, self.total = device_mem_info(); self.min_free = self.free0

    def run(self):
        while not self.stop:
            self.min_free = min(self.min_free, device_mem_info()[0]); time.sleep(0.005)


def test_configs3_rank_share_through_ft_gen(monkeypatch):
    from hifiasm_amd import synth, memplan
    from hifiasm_amd.api import Engine
    from hifiasm_amd.workloads import WORKLOADS, GENOME_SEED, READ_SEED, n_reads_of
    gs, cov, rl, err = WORKLOADS["human3G_hifi40x"][:4]
    n_all = n_reads_of("human3G_hifi40x"); n_loc = n_all // 8
    t0 = time.time()
    g = synth.make_genome(gs, seed=GENOME_SEED)
    rs = synth.make_reads(g, n_loc, rl, err, seed=READ_SEED, rid0=0, want_codes=False)
    del g
    t_gen = time.time() - t0
    assert rs.total_bases > 14.5e9
    res = {}
    for tag, passes in (("auto", None), ("as_a_rank", 6)):
        if passes is not None:
            monkeypatch.setenv("HAO_FT_PASSES", str(passes))
        poll = _PeakPoll(); poll.start()
        e = Engine(0)
        try:
            e.set_readset(rs)
            t1 = time.time(); hom = e.ha_ft_gen(); dt = time.time() - t1
            poll.stop = True; poll.join()
            k, v = e.ft_table()
            res[tag] = dict(passes=e.ft_passes(), hom=hom, hist=e.hist(0), keys=k.copy(), vals=v.copy(), wall_s=round(dt, 2), peak_gb=round((poll.total - poll.min_free) / 1e9, 1),
                            before_gb=round((poll.total - poll.free0) / 1e9, 1), total_gb=round(poll.total / 1e9, 1))
        finally:
            poll.stop = True
            e.close()
    a, d = res["auto"], res["as_a_rank"]
    occ_ = int((a["hist"].astype(np.int64) * np.arange(a["hist"].size)).sum())
    assert a["passes"] == memplan.ft_passes(occ_, (a["total_gb"] - a["before_gb"]) * 1e9, False) and d["passes"] == 6 and a["passes"] == max(1, -(-occ_ // (1 << 32)))
    assert memplan.ft_passes(occ_, (a["total_gb"] - a["before_gb"]) * 1e9, True) >= 2      # the same share as one of eight ranks does not fit in one pass
    assert a["peak_gb"] < 0.95 * a["total_gb"] and d["peak_gb"] <= a["peak_gb"] + 1.0, (a["peak_gb"], d["peak_gb"])      # (from three passes on the peak is no longer the pass buffers: measured 149.5 GB with three and with six)
    assert a["hom"] == d["hom"] and (a["hist"] == d["hist"]).all() and a["keys"].shape == d["keys"].shape and (a["keys"] == d["keys"]).all() and (a["vals"] == d["vals"]).all()
    # exact counting: every occurrence is in exactly one run; counts saturate at 4095 only in the histogram's last bin (a random genome has no such k-mer)
    h = a["hist"].astype(np.int64)
    occ = int((h * np.arange(h.size)).sum())
    assert h[4095] == 0 and 0.9 * rs.total_bases * 0.75 < occ < rs.total_bases      # (HPC: ~0.75 runs per base of a random genome, minus k - 1 per read)
    plan = memplan.rank_plan(float(gs) * cov / 8, n_loc, 1, 0.02873, 11_900 * cov / 30.0, float(gs))      # this device's share as a world of one
    out = {"workload": "human3G_hifi40x reads [0, 1e6)", "bases": rs.total_bases, "kmer_occurrences": occ, "distinct_kmers": int(h.sum()), "gen_s": round(t_gen, 1),
           "auto": {q: a[q] for q in ("passes", "wall_s", "peak_gb", "before_gb", "total_gb", "hom")}, "as_a_rank": {q: d[q] for q in ("passes", "wall_s", "peak_gb")},
           "plan_passes": plan["passes_ft"], "plan_ft_gen_gb": round(plan["ft_gen"] / 1e9, 1)}
    print("[rank share] " + json.dumps(out))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "rankshare_ft.json"), "w").write(json.dumps(out) + "\n")
    except OSError:
        pass
