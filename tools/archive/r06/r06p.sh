cd $GRAFT_REPO_ROOT; O=gpurun_out/r06p; mkdir -p $O
for p in 1 2 3 4 6; do echo "== HAO_FT_PASSES=$p"; HAO_FT_PASSES=$p timeout 300 python tools/ft_time.py chr1_250M_hifi30x 3 2>&1 | grep -A2 "call [12]" | cut -c1-400; done > $O/ft_passes.txt 2>&1; cat $O/ft_passes.txt
