#!/bin/bash
# round 3, first GPU call: the suite, the bench line, and counters of every kernel of a Cm trial (before the rewrites)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" ; tail -3 $O/pytest.log
python tools/pmc_all.py $O/pmc_pipe -- python3 $GRAFT_REPO_ROOT/tools/pipeline_time.py Cm 1234 4 > $O/pmc_pipeline_Cm.json 2> $O/pmc_pipeline_Cm.err
echo "pmc_all done"
STOCS_DEBUG_BASES=1 python tools/pipeline_time.py Cm 1234 6 > $O/pipe.json 2> $O/pipe.err
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; head -c 300 $O/bench.json
