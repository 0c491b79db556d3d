#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 300 python tools/lcp_ab.py C5 4 31,39,48 > $O/ab_C5_48.json 2> $O/ab.err; cat $O/ab_C5_48.json
timeout -k 10 300 python tools/lcp_ab.py dense 4 31,39,48 > $O/ab_dense_48.json 2>> $O/ab.err; cat $O/ab_dense_48.json
