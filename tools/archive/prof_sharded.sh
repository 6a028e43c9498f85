cd /tmp && export TMPDIR=/tmp
HAO_BENCH_FORCE_SHARDED=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_shard -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_shard.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_shard -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if 'nccl' in n.lower() or 'rccl' in n.lower() or 'AllGather' in n or 'SendRecv' in n or 'AllReduce' in n or 'Broadcast' in n:
        print(f"{n[:80]:80s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
tail -1 gpurun_out/prof_shard.log | cut -c1-200
