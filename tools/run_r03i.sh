#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
timeout -k 10 300 python tools/frame_latency.py 8 > $O/frame_latency.json 2> $O/frame.err; echo "frame rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03i/frame_latency.json"))
for k,v in d.items(): print(k, v["scene_points"], {a:round(b,3) for a,b in v["median_ms"].items()}, v["best_lcp"][:3])
PY
