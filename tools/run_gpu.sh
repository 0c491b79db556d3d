#!/bin/bash
# GPU-box helper used during development: tests, then the Cm trial-stream timing with the per-stage timers on.
# usage (from the repo root, through gpurun): bash tools/run_gpu.sh <tag> [pytest-selection]
set -e
TAG=${1:-dev}
SEL=${2:-tests}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python -m pytest $SEL -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1 || { tail -40 gpurun_out/$TAG/pytest.log; exit 1; }
tail -3 gpurun_out/$TAG/pytest.log
STOCS_DEBUG_TIMING=1 python tools/pipeline_time.py Cm 1234 6 > gpurun_out/$TAG/pipe.json 2> gpurun_out/$TAG/pipe_timing.log
python tools/pipeline_time.py Cm 1234 8 > gpurun_out/$TAG/pipe_untimed.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/pipe_stats -- python3 $R/tools/pipeline_time.py Cm 1234 5 > $R/gpurun_out/$TAG/pipe_stats.log 2>&1
echo done
