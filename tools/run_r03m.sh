#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/pipeline_time.py Cm 1234 10 > $O/log 2>&1; echo "stats rc $?"
cd $R
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r03m/stats/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:40]:
    print("%-90s calls %5s avg %9.1f us total %8.2f ms %5s%%"%(r["Name"][:90],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6,r["Percentage"]))
PY
