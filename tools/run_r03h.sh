#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
for k in 1 2 3 4; do
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-pmc > $O/benchfix_$k.json 2> $O/benchfix_$k.err; echo "bench $k rc $?"
done
python - <<'PY'
import json
for k in (1,2,3,4):
    d=json.load(open("gpurun_out/r03h/benchfix_%d.json"%k)); p=d["pipeline"]
    print(k, round(d["value"]/1e6,2), p["runs_with_a_phase_over_10x_its_median"], [round(max(x["sample_ms"],x["congruent_ms"],x["transforms_ms"],x["verify_ms"]),2) for x in p["runs"]], round(p["steady_state_poses_per_s_phases_2_4"]/1e6,2))
PY
grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat
