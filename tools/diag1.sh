set -x
for wl in bacterial5M_hifi30x bacterial5M_hifi30x_repeat; do
  HAO_DBG_DP_STATS=1 python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep -E "^\[dp\]|stage_ms" | sed 's/.*"stage_ms"/stage_ms/' | cut -c1-700 | tail -2
  HAO_DBG_DP_NOTAIL=1 python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep -E "stage_ms" | sed 's/.*"stage_ms"/NOTAIL stage_ms/' | cut -c1-700
done
