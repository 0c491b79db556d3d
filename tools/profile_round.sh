#!/bin/bash
# Reproduces the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r02
# Separate passes: kernel-trace stats, then one --pmc pass per counter group (tools/pmc.py; never combined with
# sys/hip/hsa traces).  Everything lands in gpurun_out/<tag>/; tools/collect_profiles.py copies the summaries to profiles/.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT/stats $OUT/pipe
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-pipeline --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 20 --warmup 3 > $OUT/stats/log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipe -- python3 $R/tools/pipeline_time.py Cm 1234 6 > $OUT/pipe/log 2>&1
python3 $R/tools/pmc.py join_count_kernel $OUT/pmc_join -- python3 $R/tools/pipeline_time.py Cm 1234 4 > $OUT/join_pmc.json 2> $OUT/join_pmc.err
python3 $R/tools/pmc.py lcp_coop $OUT/pmc_lcp -- $B --steps 3 --warmup 1 > $OUT/lcp_pmc.json 2> $OUT/lcp_pmc.err
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 bench.py --workload C5 --steps 40 --warmup 3 --no-pipeline --cpu-seconds 6 > $OUT/bench_C5.json 2> $OUT/bench_C5.err
python3 tools/frame_latency.py 8 > $OUT/frame_latency.json
python3 tools/sweep.py > $OUT/sweep.json
python3 tools/trials.py --trials 64 --seed 3 > $OUT/trials64_s1.json
python3 tools/trials.py --trials 64 --seed 3 --streams 8 > $OUT/trials64_s8.json
python3 tools/pipeline_time.py Cm 1234 8 > $OUT/pipeline_Cm.json
echo "profiles written under $OUT"
