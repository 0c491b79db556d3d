#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" ; tail -3 $O/pytest.log
timeout -k 10 900 python bench.py --workload C5 --steps 40 --warmup 3 --no-pipeline --cpu-seconds 6 > $O/bench_C5.json 2> $O/bench_C5.err; echo "bench C5 rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03g/bench_C5.json"))
r=d["roofline"]
print(d["value"], r["kernel_ms"], r["frac"], r["traffic"], r["traffic_over_algorithmic"], r["binding"]["bound"], r["binding"]["all"])
PY
