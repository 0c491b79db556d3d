#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_examples.py tests/test_driver_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout -k 10 300 python tools/join_ablate.py > $O/join_ablate2.json 2> $O/join.err; cat $O/join_ablate2.json
timeout -k 10 200 python tools/pipeline_time.py Cm 1234 8 > $O/pipe.json 2> $O/pipe.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03j/pipe.json"))
print([round(r["t_congruent_ms"],3) for r in d["runs"]], [round(r["poses_per_s_phases_2_4"]/1e6,2) for r in d["runs"]])
PY
