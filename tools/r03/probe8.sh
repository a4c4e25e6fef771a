#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p8
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -X faulthandler -m pytest tests -x -q -s -m gpu > $OUT/full.txt 2>&1
echo "rc=$?"
grep -v "^  File" $OUT/full.txt | tail -40 | cut -c1-400
