"""hao_attach: a second batch context over one engine's reads and index.  Two host threads, one per context, run the halves of a pass concurrently;
every read must still equal the oracle's result, a view must follow the owner's index rebuilds, and calls that change reads or index are refused on it."""
import threading

import numpy as np
import pytest

from helpers import scenario_reads, scenario_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["rr", "hifi_15k"])
def test_two_contexts_run_a_pass_together(name):
    from hifiasm_amd.api import Engine, HaoError
    rs, okw = scenario_reads(name)
    o = scenario_oracle(name)
    e = Engine(0, **okw)
    e.set_readset(rs); e.ha_ft_gen(); e.ha_pt_gen()
    v = e.attach()
    step = max(1, rs.n // 6)
    ranges = [(lo, min(rs.n, lo + step)) for lo in range(0, rs.n, step)]
    for round_ in range(2):                      # second round: after the owner rebuilt its index (the view takes the new buffers by itself)
        got, errs = {}, []

        def run(eng, rr):
            try:
                for lo, hi in rr:
                    eng.overlap_batch(lo, hi)
                    for r in range(lo, hi):
                        got[r] = tuple(np.array(x) for x in eng.h_ec_lchain(r))      # copies: the next batch reuses the host buffers
            except Exception as ex:              # noqa: BLE001 - reported by the main thread
                errs.append(ex)

        th = [threading.Thread(target=run, args=(e, ranges[0::2])), threading.Thread(target=run, args=(v, ranges[1::2]))]
        [t.start() for t in th]; [t.join() for t in th]
        assert not errs, errs
        bad = []
        for r in range(rs.n):                    # (the oracle is not thread-safe: compared here)
            ol, fc, fo, cl = got[r]; ool, ofc, ofo, ocl = o.lchain(r)
            if not (ol.shape == ool.shape and (ol == ool).all() and fc.shape == ofc.shape and (fc == ofc).all() and (fo == ofo).all() and cl.shape == ocl.shape and (cl == ocl).all()):
                bad.append(r)
        assert not bad, f"round {round_}: {len(bad)}/{rs.n} reads differ, first {bad[:5]}"
        e.ha_pt_gen()
    with pytest.raises(HaoError):
        v.ha_pt_gen()                            # the index belongs to the owner
    with pytest.raises(HaoError):
        v.set_readset(rs)
    # streaming delivery through a view
    slot = v.overlap_batch_async(0, min(rs.n, 64))
    d = v.deliver_wait(slot)
    for r in range(0, min(rs.n, 64), 7):
        ool, ofc, ofo, ocl = o.lchain(r)
        ol, fc, fo, cl = v.delivered_read(d, r)
        assert ol.shape == ool.shape and (ol == ool).all() and cl.shape == ocl.shape and (cl == ocl).all()
    v.close(); e.close()
