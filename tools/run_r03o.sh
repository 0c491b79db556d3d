#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 900 python tools/big_model_check.py 18000 3 > $O/big.json 2> $O/big.err; echo "rc $?"; tail -5 $O/big.err; cat $O/big.json
