# kernel resource usage (SGPR / VGPR / scratch / LDS / occupancy) of every kernel in libhao.so's translation unit, from the compiler (no GPU needed)
# usage: bash tools/kres.sh [kernel name pattern]
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -I$R/include -I$R/hifiasm_amd/csrc --cuda-device-only -c $R/hifiasm_amd/csrc/hao_capi.hip -o /tmp/hao_dev.o -Rpass-analysis=kernel-resource-usage 2> /tmp/kres.txt
python3 - "$1" <<'PY'
import re,sys
pat=sys.argv[1] if len(sys.argv)>1 else ""
cur=None; rows={}
for ln in open('/tmp/kres.txt'):
    m=re.search(r"Function Name: (\S+)",ln)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (\d+)",ln)
    if m and cur: rows[cur][m.group(1)]=int(m.group(2))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    name=re.sub(r"\(.*","",name).replace("void ","")
    if pat and pat not in name: continue
    if name.startswith("rocprim::"): continue
    print(f"{name[:70]:70s} sgpr {v.get('TotalSGPRs',0):4d} vgpr {v.get('VGPRs',0):4d} scratch {v.get('ScratchSize [bytes/lane]',0):5d} lds {v.get('LDS Size [bytes/block]',0):6d} occ {v.get('Occupancy [waves/SIMD]',0)} spill s{v.get('SGPRs Spill',0)} v{v.get('VGPRs Spill',0)}")
PY
