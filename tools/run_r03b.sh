#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py tests/test_examples.py tests/test_driver_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" ; tail -5 $O/pytest.log
timeout -k 10 200 python tools/pipeline_time.py Cm 1234 8 > $O/pipe.json 2> $O/pipe.err; echo "pipe rc $?"
STOCS_CONGRUENT_RADIX=1 timeout -k 10 200 python tools/pipeline_time.py Cm 1234 8 > $O/pipe_radix.json 2> $O/pipe_radix.err
python - <<'PY'
import json
for f in ("pipe","pipe_radix"):
    d=json.load(open("gpurun_out/r03b/%s.json"%f))
    print(f, [round(r["t_congruent_ms"],3) for r in d["runs"]], [r["quads"] for r in d["runs"]][:3])
PY
timeout -k 10 120 tools/bin/ta_microbench 2000 > $O/ta_microbench.json 2> $O/ta.err; echo "ta rc $?"
timeout -k 10 300 python tools/pmc_all.py $O/pmc_pipe -- python3 $GRAFT_REPO_ROOT/tools/pipeline_time.py Cm 1234 4 > $O/pmc_pipeline_Cm.json 2> $O/pmc_pipeline_Cm.err
