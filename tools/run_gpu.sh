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

if [ -n "$PMC" ]; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pipe_sq -- python3 $R/tools/pipeline_time.py Cm 1234 3 > $R/gpurun_out/$TAG/pipe_sq.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU TA_TA_BUSY_sum TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/pipe_busy -- python3 $R/tools/pipeline_time.py Cm 1234 3 > $R/gpurun_out/$TAG/pipe_busy.log 2>&1
  echo pmc done
fi
