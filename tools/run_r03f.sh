#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_lcp_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout -k 10 300 python tools/lcp_ab.py C5 4 31,39 > $O/ab_C5_b.json 2> $O/ab.err; cat $O/ab_C5_b.json
timeout -k 10 300 python tools/lcp_ab.py dense 4 31,39 > $O/ab_dense_b.json 2>> $O/ab.err; cat $O/ab_dense_b.json
