#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p4
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_hip_stats.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/r03/probe4.py > $OUT/ops.txt 2>&1
head -150 $OUT/ops.txt | cut -c1-330
