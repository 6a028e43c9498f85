"""f3 end to end on the device (hao_window_ed_grid): pairs/s INCLUDING task generation on BASELINE configs[1] (10 000 reads of 15 kb) - every overlap of every read of
one all-reads batch on the reference's window grid (WINDOW = 375), one threshold per call.  Prints one JSON line.  usage: python tools/bench_ed_resident.py [thre] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from hifiasm_amd import workloads
    from hifiasm_amd.api import Engine
    thre = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rs = workloads.workload_reads("bacterial5M_hifi30x")
    e = Engine(0); e.set_readset(rs); e.ha_ft_gen(); e.ha_pt_gen()
    e.overlap_batch(0, rs.n)
    n = e.window_ed_grid(375, thre)      # warm-up (allocations)
    ts = []
    for _ in range(reps):
        t0 = time.time(); n = e.window_ed_grid(375, thre); ts.append(time.time() - t0)
    t, r = e.fetch_ed_grid(min(n, 1_000_000))
    ok = int((r[:, 0] != 2**31 - 1).sum())
    print(json.dumps({"workload": "bacterial5M_hifi30x", "reads": int(rs.n), "overlaps": e.batch_totals()["overlaps"], "window": 375, "thre": thre, "pairs": n,
                      "ms_per_call_best": round(min(ts) * 1e3, 3), "ms_per_call_all": [round(x * 1e3, 3) for x in ts], "pairs_per_s": round(n / min(ts)),
                      "within_thre_of_first_million": ok, "what": "task generation on the device from ol->list + distance-only window alignment; nothing crosses the host but two totals"}))
    e.close()


if __name__ == "__main__":
    main()
