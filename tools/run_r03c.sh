#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_lcp_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 200 python tools/lcp_table_ab.py Cm 8 0 1,0 > $O/table_ab_Cm.json 2> $O/table_ab.err; cat $O/table_ab_Cm.json
timeout -k 10 200 python tools/lcp_table_ab.py Cm 8 8192 1,0 > $O/table_ab_Cm8k.json 2>> $O/table_ab.err; cat $O/table_ab_Cm8k.json
