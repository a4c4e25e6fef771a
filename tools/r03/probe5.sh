#!/bin/bash
# r03p5: the whole gpu suite after the training-graph changes, then the steady-state table of the training iteration
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p5
rm -rf $OUT; mkdir -p $OUT
cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) 2>&1 | tee $OUT/pytest.txt
bash tools/r03/probe2.sh convs
