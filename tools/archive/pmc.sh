# HBM traffic per kernel: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE do not fit one pass) over one bench step.
# usage (on the GPU box): bash tools/pmc.sh <tag> [bench args]  -> gpurun_out/pmc_<tag>/{fetch,write}/...counter_collection.csv + pmc_<tag>.json
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $out.$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $out "$@" > gpurun_out/pmc_$tag.json
python - <<PY
import json
d=json.load(open("gpurun_out/pmc_$tag.json"))
ks=sorted(d["kernels"].items(), key=lambda kv:-kv[1]["hbm_bytes_per_launch"]*kv[1]["launches"])[:12]
for k,v in ks: print(f"{k[:50]:50s} launches {v['launches']:3d}  hbm MB/launch {v['hbm_bytes_per_launch']/1e6:10.1f}")
PY
