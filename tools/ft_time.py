"""ha_ft_gen's wall clock, stage by stage: GPU time between the engine's stage marks (hao_stage_times) and the HOST's wall clock between the same marks ("host_ft_...":
allocation, rocPRIM scratch, the per-read host loops).  usage: python tools/ft_time.py [workload] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hifiasm_amd.api import Engine
from hifiasm_amd.workloads import workload_reads

wl = sys.argv[1] if len(sys.argv) > 1 else "chr1_250M_hifi30x"
rs = workload_reads(wl)
e = Engine(0); e.set_readset(rs)
for call in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    t0 = time.time(); hom = e.ha_ft_gen(); dt = time.time() - t0
    st = e.stage_times()
    gpu = {k: round(v, 1) for k, v in st if not k.startswith("host_")}; host = {k[5:]: round(v, 1) for k, v in st if k.startswith("host_")}
    print(f"[ft_time] {wl} call {call}: wall {dt * 1e3:.0f} ms, hom peak {hom}, passes {e.ft_passes()}\n  gpu ms  {gpu} = {sum(gpu.values()):.0f}\n  host ms {host} = {sum(host.values()):.0f}", flush=True)
e.close()
