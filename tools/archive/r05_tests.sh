# round 5: the new GPU tests, group by group (own time limit each, --durations).  usage: bash tools/r05_tests.sh <tag> [groups: passes rank dropin heavy light all]
tag=$1; shift; what="${*:-passes rank dropin heavy light}"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
has() { case " $what " in *" $1 "*) return 0;; esac; return 1; }
run() { name=$1; lim=$2; shift; shift; t0=$(date +%s); timeout $lim python -m pytest "$@" -q -m gpu --durations=6 -rxXfs > $O/pytest_$name.log 2>&1; rc=$?; echo "[$name] rc=$rc $(( $(date +%s) - t0 )) s: $(tail -1 $O/pytest_$name.log)"; grep -E '^(FAILED|ERROR)' $O/pytest_$name.log | head -5; }
has passes && run passes 300 tests/test_gpu_tables.py tests/test_gpu_sharded.py -k "passes"
has rank && run rank 600 tests/test_gpu_zz_rankshare.py -s
has dropin && run dropin 600 tests/test_gpu_dropin.py -s
has heavy && run heavy 900 tests/test_gpu_fuzz.py -k "repeat_dense"
has light && run light 900 tests/test_gpu_fuzz.py -k "random_workload"
has all && run all 1500 tests
grep -h '^\[rank share\]\|^\[dropin configs1\]' $O/pytest_*.log | cut -c1-1500
