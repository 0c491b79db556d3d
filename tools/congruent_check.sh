#!/bin/bash
# The GPU tests of phases 1-4 plus the step record of a Cm trial (tools/pipeline_time.py): the loop used while working on the congruent phase.  Run on the GPU box.
cd $GRAFT_REPO_ROOT
O=gpurun_out/congruent_check; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py tests/test_examples.py tests/test_driver_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/pipeline_time.py Cm 1234 12 > $O/pipe.json 2> $O/pipe.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/congruent_check/pipe.json"))
print([round(r["t_congruent_ms"],3) for r in d["runs"]], [round(r["poses_per_s_phases_2_4"]/1e6,2) for r in d["runs"]])
for c,v in d["steps_ms_median"].items():
    print(c)
    for lab,ms in v: print("   %-60s %.4f"%(lab,ms))
import numpy as np
for k in ("t_sample_ms","t_congruent_ms","t_transforms_ms","t_verify_ms"): print(k, np.median([r[k] for r in d["runs"][2:]]))
PY
