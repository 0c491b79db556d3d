#!/bin/bash
# Reproduces the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r03
# Separate passes: kernel-trace stats, then one --pmc pass per counter group (tools/pmc.py / tools/pmc_all.py; never combined with
# sys/hip/hsa traces).  Everything lands in gpurun_out/<tag>/; tools/collect_profiles.py copies the summaries to profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT/stats
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-pipeline --no-pmc"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 20 --warmup 3 > $OUT/stats/log 2>&1; echo "stats rc $?"
timeout -k 10 400 python3 $R/tools/pmc_all.py $OUT/pmc_pipe -- python3 $R/tools/pipeline_time.py Cm 1234 4 > $OUT/pmc_pipeline_Cm.json 2> $OUT/pmc_pipeline_Cm.err; echo "pmc_all rc $?"
timeout -k 10 300 python3 $R/tools/pmc.py lcp_coop $OUT/pmc_lcp -- $B --steps 3 --warmup 1 > $OUT/lcp_pmc.json 2> $OUT/lcp_pmc.err; echo "pmc lcp rc $?"
cd $R
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
timeout -k 10 900 python3 bench.py --workload C5 --steps 40 --warmup 3 --no-pipeline --cpu-seconds 6 > $OUT/bench_C5.json 2> $OUT/bench_C5.err; echo "bench C5 rc $?"
timeout -k 10 300 python3 tools/frame_latency.py 8 > $OUT/frame_latency.json 2> $OUT/frame.err; echo "frame rc $?"
timeout -k 10 600 python3 tools/sweep.py > $OUT/sweep.json 2> $OUT/sweep.err; echo "sweep rc $?"
timeout -k 10 300 python3 tools/trials.py --trials 64 --seed 3 > $OUT/trials64_s1.json 2> $OUT/trials.err; echo "trials rc $?"
timeout -k 10 300 python3 tools/trials.py --trials 64 --seed 3 --streams 8 > $OUT/trials64_s8.json 2>> $OUT/trials.err
timeout -k 10 300 python3 tools/pipeline_time.py Cm 1234 10 > $OUT/pipeline_Cm.json 2> $OUT/pipe.err; echo "pipeline rc $?"
timeout -k 10 300 python3 tools/stall_watch.py 64 1234 0 > $OUT/stall_watch.json 2> $OUT/stall.err; echo "stall rc $?"
timeout -k 10 300 python3 tools/percall_time.py > $OUT/percall.json 2> $OUT/percall.err; echo "percall rc $?"
echo "profiles written under $OUT"
