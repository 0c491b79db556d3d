#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-pmc > $O/bench_quick.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03f/bench_quick.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["pipeline"]["steady_state_poses_per_s_phases_2_4"], d["pipeline"]["runs_with_a_phase_over_10x_its_median"])
for x in d["pipeline"]["runs"]: print(x["warmup"], x["bases"], x["candidates"], round(x["sample_ms"],3), round(x["congruent_ms"],3), round(x["transforms_ms"],3), round(x["verify_ms"],3))
PY
