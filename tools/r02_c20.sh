cd $GRAFT_REPO_ROOT; O=gpurun_out/c20; mkdir -p $O
for m in 0 1 2; do for cpu in 0 64; do HAO_DBG_DLTIME=1 HAO_ARENA_NUMA=$m timeout 600 taskset -c $cpu-$((cpu+63)) python bench.py --cpu-baseline none --steps 2 > $O/b_${m}_$cpu.json 2> $O/b_${m}_$cpu.err; grep "arena 0" $O/b_${m}_$cpu.err | head -1; python - <<PY
import json
d=json.loads(open("$O/b_${m}_$cpu.json").read().strip().splitlines()[-1]); b=d['boundary']
print("mode $m cpus $cpu", d['ms_per_step'], round(d['value_boundary']/1e6,1), b['ms_per_step'], b['copy_ms_per_step'], b['copy_gb_per_s'])
PY
done; done
cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c | head
