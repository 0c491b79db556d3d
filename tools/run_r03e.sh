#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 300 python tools/stall_watch.py 64 1234 0 > $O/stall_distinct.json 2> $O/stall.err; echo "stall rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03e/stall_distinct.json"))
print("stalled:", [(x["trial"], x["seed"], x["quads"], [round(v,2) for v in x["ms"]]) for x in d["stalled_trials"]])
mx=0
for s,q,ms in d["all_trials_seed_quads_congruent_ms"]:
    print(s, q, ms, "<-- new max quads" if q>mx else "")
    mx=max(mx,q)
PY
timeout -k 10 600 python tools/lcp_ablate.py C5 3 > $O/ablate_C5.json 2> $O/ablate.err; echo "ablate rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03e/ablate_C5.json"))
for c in d["cases"]: print("%3d %-75s %.4f ms" % (c["ablate"], c["what"], c["ms_median"]))
PY
timeout -k 10 300 python tools/lcp_ab.py C5 3 31,34,35 > $O/ab_C5_unr.json 2>> $O/ablate.err; cat $O/ab_C5_unr.json
