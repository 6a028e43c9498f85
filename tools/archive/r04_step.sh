# usage: bash tools/r04_step.sh <tag> "<pytest targets>" [legs of tools/r04_run.sh ...]   (targeted tests first, stop on failure; then measurement legs)
tag=$1; tests=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
if [ -n "$tests" ]; then timeout 900 python -m pytest $tests -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > $O/pytest.log; cat $O/pytest.log; fi
if [ $# -gt 0 ]; then bash tools/r04_run.sh $tag "$@"; fi
