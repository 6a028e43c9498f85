# last call of round 4: the seed kernels' slot-matched scatter pass - bench (full line), kernel trace, the whole -m gpu suite, the small workloads, the seed counters
tag=${1:-r04fin}
export CPU_BASELINE=sample BENCH_ARGS="--steps 20 --warmup 5" TEST_TIMEOUT=${TEST_TIMEOUT:-230}
for leg in bench prof tests benchall seedctr; do echo "== $leg $(date +%s)"; bash tools/r04_run.sh $tag $leg; done
