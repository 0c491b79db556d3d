set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1 || { tail -30 gpurun_out/r02a/pytest.log; exit 1; }
tail -3 gpurun_out/r02a/pytest.log
STOCS_DEBUG_TIMING=1 python tools/pipeline_time.py Cm 1234 6 > gpurun_out/r02a/pipe.json 2> gpurun_out/r02a/pipe_timing.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02a/pipe_stats -- python3 $R/tools/pipeline_time.py Cm 1234 5 > $R/gpurun_out/r02a/pipe_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r02a/pipe_sq -- python3 $R/tools/pipeline_time.py Cm 1234 3 > $R/gpurun_out/r02a/pipe_sq.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/r02a/pipe_tcp -- python3 $R/tools/pipeline_time.py Cm 1234 3 > $R/gpurun_out/r02a/pipe_tcp.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU TA_TA_BUSY_sum TCC_REQ_sum TCC_READ_sum TCC_BUSY_sum TCC_CYCLE_sum --kernel-trace --output-format csv -d $R/gpurun_out/r02a/pipe_busy -- python3 $R/tools/pipeline_time.py Cm 1234 3 > $R/gpurun_out/r02a/pipe_busy.log 2>&1 || echo "busy pass failed"
echo done
