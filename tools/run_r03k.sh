#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 200 python tools/pipeline_time.py Cm 1234 12 > $O/pipe.json 2> $O/pipe.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03k/pipe.json"))
print([round(r["t_congruent_ms"],3) for r in d["runs"]], [round(r["poses_per_s_phases_2_4"]/1e6,2) for r in d["runs"]])
for c,v in d["steps_ms_median"].items():
    print(c)
    for lab,ms in v: print("   %-60s %.4f"%(lab,ms))
import numpy as np
for k in ("t_sample_ms","t_congruent_ms","t_transforms_ms","t_verify_ms"): print(k, np.median([r[k] for r in d["runs"][2:]]))
PY
