#!/bin/bash
# Reproduces the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r01
# Separate passes: kernel-trace stats, then one --pmc pass per counter group (never combined with
# sys/hip/hsa traces).  Summaries land in gpurun_out/; tools/collect_profiles.py copies them to profiles/.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/tcp $OUT/ta $OUT/pipe
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-pipeline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 10 --warmup 2 > $OUT/stats/log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B --steps 3 --warmup 1 > $OUT/fetch/log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B --steps 3 --warmup 1 > $OUT/write/log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq -- $B --steps 2 --warmup 1 > $OUT/sq/log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcp -- $B --steps 2 --warmup 1 > $OUT/tcp/log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum --kernel-trace --output-format csv -d $OUT/ta -- $B --steps 2 --warmup 1 > $OUT/ta/log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipe -- python3 $R/tools/pipeline_time.py Cm 1234 5 > $OUT/pipe/log 2>&1
cd $R && python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "profiles written under $OUT"
