#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p7
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -X faulthandler -m pytest tests/test_hip_nets.py -x -q -s -m gpu > $OUT/nets.txt 2>&1
echo "rc=$?"
grep -v "^  File" $OUT/nets.txt | head -60 | cut -c1-300
timeout 600 python -m pytest tests/test_nn_ops.py tests/test_hip_stats.py -q -m gpu 2>&1 | tail -5
