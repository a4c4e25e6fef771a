#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03p9
rm -rf $OUT; mkdir -p $OUT
cd $R
export AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1
timeout 1500 python -X faulthandler -m pytest tests/test_hip_agent.py tests/test_hip_evaluate.py tests/test_hip_filters.py tests/test_hip_nets.py -x -q -s -m gpu > $OUT/full.txt 2>&1
echo "rc=$?"
grep -v "Extension modules" $OUT/full.txt | tail -60 | cut -c1-300
