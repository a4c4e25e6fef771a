#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p6
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -X faulthandler -m pytest tests -x -v -m gpu > $OUT/pytest_full.txt 2>&1
echo "rc=$?"
grep -n "Fatal\|Aborted\|HSA\|hip\|tests/.*line\|PASSED\|FAILED" $OUT/pytest_full.txt | tail -30 | cut -c1-220
