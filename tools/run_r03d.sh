#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" ; tail -3 $O/pytest.log
timeout -k 10 300 python tools/stall_watch.py 200 > $O/stall_watch.json 2> $O/stall.err; echo "stall rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03d/stall_watch.json"))
print({k:d[k] for k in d if k not in ("stalled_trials","example_steps_of_a_normal_trial")}, "stalled:", len(d["stalled_trials"]))
for x in d["stalled_trials"][:4]: print(x["trial"], x["seed"], x["ms"], x["steps"])
print(d["example_steps_of_a_normal_trial"])
PY
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; head -c 400 $O/bench.json
