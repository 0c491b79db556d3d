#!/bin/bash
# Cell edge (eps / 1, 2, 4) against scene density with dominance-pruned lists: which edge wins where (thresholds of ctx.hip build_grid_levels).
cd $GRAFT_REPO_ROOT
O=gpurun_out/prune_layout; mkdir -p $O; : > $O/sweep.txt
for cfg in "20000 5000 0.005" "35000 8000 0.004" "50000 12500 0.0032" "65000 16000 0.0028" "100000 25000 0.0023" "140000 35000 0.0019" "200000 50000 0.0016"; do
  for d in 1 2 4; do
    STOCS_GRID_PRUNE=1 STOCS_GRID_DIV=$d STOCS_DEBUG_TIMING=1 timeout -k 10 200 python3 tools/layout_sweep.py $cfg 16384 2> $O/err.txt >> $O/sweep.txt || exit 1
    grep "stocs grid" $O/err.txt | tail -1 >> $O/sweep.txt
  done
done
cat $O/sweep.txt
