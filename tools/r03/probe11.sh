#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p11
rm -rf $OUT; mkdir -p $OUT
cd $R
for f in test_hip_agent test_hip_evaluate; do
  timeout 900 python -X faulthandler -m pytest tests/$f.py tests/test_hip_nets.py -x -v -m gpu > $OUT/$f.txt 2>&1
  echo "$f + nets: rc=$?"
  grep "PASSED\|FAILED\|Fatal\|fault" $OUT/$f.txt | tail -4 | cut -c1-200
done
