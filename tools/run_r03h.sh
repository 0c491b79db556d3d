#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
python - <<'PY'
import numpy, os
try:
    from threadpoolctl import threadpool_info
    print([(d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()])
except Exception as e: print("threadpoolctl", e)
print("sched_getaffinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
PY
grep -E "nr_throttled|throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null
run() { name=$1; shift
  for k in 1 2 3 4 5 6 7 8 9 10; do env "$@" timeout -k 10 120 python tools/stall_watch.py 16 1234 0 > $O/sw_${name}_$k.json 2> $O/sw_${name}_$k.err; done
  python - "$name" <<'PY'
import json,sys
name=sys.argv[1]; n=0; st=[]
for k in range(1,11):
    try:
        d=json.load(open("gpurun_out/r03h/sw_%s_%d.json"%(name,k))); n+=1
        mx=max(max(x[2] for x in d["all_trials_seed_quads_congruent_ms"]), max(d["max_ms"]))
        if mx>10: st.append((k, round(mx,1)))
    except Exception as e: pass
print(name, "processes", n, "with a phase over 10 ms:", st)
PY
}
run cpuwork_default SW_CPUWORK=1
run cpuwork_1thread SW_CPUWORK=1 OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1
grep -E "nr_throttled|throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null
