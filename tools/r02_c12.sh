cd $GRAFT_REPO_ROOT; O=gpurun_out/c12; mkdir -p $O
HAO_DBG_DLTIME=1 timeout 600 python bench.py --cpu-baseline none --steps 2 > $O/bench.json 2> $O/bench.err; grep -E "batch\]|deliver\]" $O/bench.err | tail -12
