# per-kernel SQ counters of one bench step.  usage: bash tools/r02_counters.sh <tag> [bench args]
tag=$1; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python $R/bench.py --cpu-baseline none --no-boundary --steps 1 --warmup 0 "$@" > $O/sq.log 2>&1
cd $R && python tools/kernel_counters.py $O/sq; find $O/sq -name "*.csv" -size +30M -delete
