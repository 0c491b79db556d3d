#!/bin/bash
# C5 bench at cell edge eps/1, eps/2, eps/4 (STOCS_GRID_DIV): traffic against step time (profiles/r03_C5_grid_div_sweep.json).  Run on the GPU box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/c5_div; mkdir -p $O
for d in 1 2 4; do
  STOCS_GRID_DIV=$d STOCS_DEBUG_TIMING=1 timeout -k 10 400 python3 bench.py --workload C5 --steps 20 --warmup 2 --no-pipeline --no-cpu-baseline > $O/c5_div$d.json 2> $O/c5_div$d.err || exit 1
  grep "stocs grid" $O/c5_div$d.err | tail -1
  python - <<PY
import json
d=json.loads(open("$O/c5_div$d.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("div $d", "value", d["value"], "ms", d["ms_per_step"], "traffic", r.get("traffic"), "t/alg", r.get("traffic_over_algorithmic"), "binding", (r.get("binding") or {}).get("all"))
PY
done
