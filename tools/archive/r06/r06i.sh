# round 6: the list-major seed kernel's DBG instance: ticks of every wave's step loop, its steps and its emitting blocks per read
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O
for f in 0 1; do HAO_SEED_LDS=1 HAO_DBG_SEEDPHASE=1 HAO_DBG_SEEDFLAGS=$f timeout 300 python bench.py --workload chr1_250M_hifi30x --cpu-baseline none --no-variants --no-boundary --steps 2 --warmup 1 > $O/f$f.json 2> $O/f$f.err; echo "flags $f"; grep "seed lds" $O/f$f.err | tail -3; done
